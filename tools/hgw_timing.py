"""Development probe for csrc/hgw.hip (-DSREC_HGW_TIMING): phase clocks (s_memtime, wave 0 of one workgroup) of the edge-GEMM
weight-gradient kernel inside eager steps of the bench workload.  usage (GPU box): python tools/hgw_timing.py"""
import ctypes, glob, importlib, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pk = os.path.join(root, 'sessionrec-pytorch_amd')
objs = [o for o in glob.glob(pk + '/csrc/*.o') if not o.endswith('hgw.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_HGW_TIMING',
                       '-c', pk + '/csrc/hgw.hip', '-o', '/tmp/hgw_tim.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_hgwtim.so',
                       '/tmp/hgw_tim.o'] + objs)
L = importlib.import_module('sessionrec-pytorch_amd._lib')
L.LIB_PATH = '/tmp/libsrec_hgwtim.so'
import bench
sys.argv = ['bench.py', '--step-only', '--no-graph', '--steps', '6', '--warmup', '2']
bench.main()
dll = ctypes.CDLL('/tmp/libsrec_hgwtim.so')
tim = (ctypes.c_ulonglong * 9)()
assert dll.srec_hgw_timing(tim) == 0
t = list(tim)
n = max(t[8], 1)
print('hg_wgrad workgroup (cycles of s_memtime at 100 MHz x 24: see r05 notes; raw counts): setup + id table %d, prologue loads + first stage %d,'
      ' epilogue %d; steady loop over %d chunks, per chunk: barrier %.0f, load issue %.0f, reads + MFMA %.0f, wait for chunk it + 1 %.0f, '
      'A pieces + stores %.0f' % (t[0], t[1], t[2], n, t[3] / n, t[4] / n, t[5] / n, t[6] / n, t[7] / n), file=sys.stderr)
