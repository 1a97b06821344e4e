#!/bin/bash
# the evidence set of a round, one GPU call:  tools/final_profiles.sh rNN  ->  gpurun_out/rNN_*  (copy what is kept into profiles/)
#   rNN_bench_default.json                       the line of a plain `python bench.py`
#   rNN_msgifsr_bf16_graph_{bench.json,kernel_stats.csv,step_trace.txt,breakdown.txt}   rocprofv3 --kernel-trace --stats of the step
#   rNN_pmc_flash_ce_bf16.json                   PMC passes of the fused scoring kernels (bench.py --kernel-only), one group per run
#   rNN_pmc_head.json                            PMC passes of the read-out head kernels on eager launches of the step
#   rNN_pmc_hg_kernels.json, rNN_pmc_gemm_gru.json   the same passes, the MSHGNN graph kernels / the bf16 GEMMs + fused GRU kernels
#   rNN_shard1_{step_trace,breakdown}.txt        the step a rank of an N-GPU job runs (1-rank communicator, collectives forced)
tag=${1:-r04}
cd /root/repo
out=/root/repo/gpurun_out
mkdir -p $out
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
bash tools/prof_bench.sh ${tag}_msgifsr_bf16_graph > /dev/null 2>&1
rm -rf /tmp/pmc_fl; bash tools/pmc_flash.sh /tmp/pmc_fl > $out/${tag}_pmc_flash.txt 2>&1
python tools/pmc_summarize.py /tmp/pmc_fl $out/${tag}_pmc_flash_ce_bf16.json 512 37484 256 bf16 > /dev/null 2>&1
rm -rf /tmp/pmc_hd; bash tools/pmc_run.sh /tmp/pmc_hd python /root/repo/bench.py --steps 2 --warmup 1 --step-only --no-graph > /dev/null 2>&1
python tools/pmc_json.py /tmp/pmc_hd $out/${tag}_pmc_head.json "rocprofv3 --pmc <group> --kernel-trace, one group per pass (tools/pmc_run.sh), python bench.py --step-only --no-graph (eager launches of the C3 step)" head_fwd_kernel head_bwd_kernel gemm_f32_group_kernel step_prep_kernel > $out/${tag}_pmc_head.txt 2>&1
note="rocprofv3 --pmc <group> --kernel-trace, one group per pass (tools/pmc_run.sh), python bench.py --step-only --no-graph (eager launches of the C3 step)"
python tools/pmc_json.py /tmp/pmc_hd $out/${tag}_pmc_hg_kernels.json "$note" hg_agg_node_kernel hg_bwd_dst_node_kernel hg_bwd_src_kernel hg_dots_kernel hg_colsum_cols_kernel hg_pre_kernel hg_drop_prep_kernel hg_drop_merge_kernel > $out/${tag}_pmc_hg.txt 2>&1
python tools/pmc_json.py /tmp/pmc_hd $out/${tag}_pmc_gemm_gru.json "$note" gemm16_nt_kernel gemm16_tn_kernel gru_fused_fwd gru_fused_bwd adam_rows_kernel adam_multi_kernel > $out/${tag}_pmc_gg.txt 2>&1
SREC_FORCE_COLLECTIVES=1 bash tools/prof_bench.sh ${tag}_shard1 --shard > /dev/null 2>&1
ls -la $out | grep ${tag}_
