/* libsrec_collate.so - C ABI of the native session -> flat batched graph builder (CPU, csrc/collate.cpp).
 *
 * Replaces the per-sample Python / DGL graph construction of the reference's collate path:
 *   src/utils/data/collate.py:29-44   seq_to_eop_multigraph   (kind 1)
 *   src/utils/data/collate.py:46-59   seq_to_shortcut_graph   (kind 2)
 *   src/utils/data/collate.py:61-85   seq_to_session_graph    (kind 0)
 *   src/utils/data/collate.py:87-217  seq_to_ccs_graph        (kind 3, `order` = K)
 *   src/utils/data/collate.py:219-256 collate_fn_factory / collate_fn_factory_ccs (dgl.batch offsets)
 * and emits the FlatBatch buffer of sessionrec-pytorch_amd/batch.py directly (int32 header of live counts followed by
 * 16-byte aligned fields: node / edge CSR per relation, item -> positions CSR, readout permutation).
 *
 * Pure CPU, re-entrant (called from DataLoader worker processes), allocates nothing the caller sees; all pointers
 * are HOST pointers owned by the caller.  The reference-side binding is the ctypes stub of INTEGRATION.md 2.
 */
#ifndef SREC_COLLATE_H
#define SREC_COLLATE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kind: 0 session graph, 1 EOP multigraph, 2 shortcut graph, 3 CCS heterograph of orders 1..order (order <= 6).
 * seqs: the B click sequences back to back (item ids); offs[B+1]: their offsets (every session non-empty).
 * caps (nullable): {B_cap, N_cap, E_cap, U_cap} for the capacity-padded layout (batch-independent offsets).
 * out[out_cap]: receives the int32 buffer; field_info[3*max_fields]: (offset, capacity, live length) per field in
 * schema order; *n_fields: fields written.
 * Returns the number of int32 written (> 0); 0 on a caller error (empty session, order out of range, a capacity
 * exceeded); -(required length) when out_cap is too small (nothing written: call again with a larger buffer). */
long srec_collate(int kind, const int64_t* seqs, const int64_t* offs, int B, int order, const int64_t* caps,
                  int32_t* out, long out_cap, int64_t* field_info, int max_fields, int* n_fields);

#ifdef __cplusplus
}
#endif
#endif
