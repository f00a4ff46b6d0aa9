/*
 * kslam_comm.h -- one process per GPU: the end-of-batch exchanges of a read-sharded batch over RCCL, behind the C ABI
 * (SURVEY.md section 8e).  Same library as kslam.h; no PyTorch anywhere: a C++ host (the reference's main, patched as
 * INTEGRATION.md shows) runs rank r of N with its own kslam_ctx on GPU r and calls these.
 *
 * What is exchanged, and why nothing else is (src/SLAM.h:194-249 with the read pairs of one batch sharded over the
 * ranks, the index replicated):
 *   kslam_comm_gather_batch     alignToDatabase's records of every shard to rank 0, in the reference's order -- one
 *                               ncclAllGather of four counts per rank, then ONE group of ncclSend / ncclRecv: every peer
 *                               sends four pieces over its own xGMI link straight into their final places on rank 0
 *                               (no merge step; kslam_export_shard_device wrote them in batch terms)
 *   kslam_comm_sharded_tail     the two batch-global steps of the tail, fed from all ranks: the insert sizes
 *                               (getMaxAllowedInsertSize, src/PairedOverlap.h:314-360: a variable-length all-gather, 4
 *                               bytes per properly paired read pair) and, with pseudo-assembly on, pseudoAssembly
 *                               (:480-582) with the ENTRIES partitioned over the ranks -- entry e on rank e mod N: an
 *                               all-to-all of 16-byte record heads to the entries' owners and one of 4-byte scores back
 *                               (one group of ncclSend / ncclRecv each; 1 / N of the stage per rank instead of all of it
 *                               on every rank); everything else of the tail is per read pair and stays on the rank that
 *                               aligned the pair
 * Failing together: when a step fails on ONE rank (an export, an allocation, a device stage that declines), a status word
 * exchanged before the next transfer makes EVERY rank return a failure for the batch -- no rank waits in a collective for
 * a peer that has left.  A transfer that itself fails half-way (RCCL / HIP error) aborts the communicator
 * (ncclCommAbort): the peers come out with an error, and the communicator is dead -- every later call returns
 * KSLAM_ERR_STATE, kslam_comm_destroy is what is left.
 * The library opens librccl.so at run time (dlopen, the first kslam_comm_* call): a single-GPU host never loads it, and
 * a process that also carries PyTorch's own copy of RCCL keeps the two apart.  Which file: the environment variable
 * KSLAM_RCCL_LIB when it is set (a path or soname; meant for a ROCm installed elsewhere and for the repository's tests, which
 * name tests/fake_rccl's stand-in there so that several ranks can share one GPU), else librccl.so.1, librccl.so,
 * /opt/rocm/lib/librccl.so.1 in that order.  The variable decides what code the product loads: a deployment that does
 * not control its environment should not run with it set.  kslam_comm_info reports the file that was opened.
 *
 * Rendezvous: rank 0 calls kslam_comm_unique_id and hands the 128 bytes to the other ranks by whatever the host has
 * (a file, a socket, MPI, torch.distributed's store); every rank then calls kslam_comm_create.
 */
#ifndef KSLAM_COMM_H_
#define KSLAM_COMM_H_
#include "kslam.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KSLAM_COMM_ID_BYTES 128 /* NCCL_UNIQUE_ID_BYTES */

typedef struct kslam_comm kslam_comm;

/* message of the calling thread's last failed kslam_comm_* call */
const char *kslam_comm_last_error(void);

/* ncclGetUniqueId */
kslam_status kslam_comm_unique_id(uint8_t id[KSLAM_COMM_ID_BYTES]);
/* ncclCommInitRank on ctx's device; collective over the `world` ranks.  The communicator works on the context's stream order:
 * every call below returns when its transfers have completed. */
kslam_status kslam_comm_create(kslam_ctx *ctx, const uint8_t id[KSLAM_COMM_ID_BYTES], int rank, int world,
                               kslam_comm **out);
void kslam_comm_destroy(kslam_comm *comm);
int kslam_comm_rank(const kslam_comm *comm);
int kslam_comm_world(const kslam_comm *comm);

/* Where every rank's four pieces land in the gathered arrays (all R1 blocks in rank order, then all R2 blocks): pure
 * arithmetic on the all-gathered counts, exported for the tests.  row1/row2: record index of rank r's R1 / R2 rows;
 * op1/op2: word index of their CIGAR words; totals[0] = records, totals[1] = CIGAR words. */
void kslam_comm_gather_plan(const kslam_shard_counts *counts, int world, uint64_t *row1, uint64_t *row2,
                            uint64_t *op1, uint64_t *op2, uint64_t totals[2]);

/* After kslam_align_resident (or any entry that leaves the results on the device) on every rank, whose batch was pairs
 * [pair_lo, pair_lo + n_local_pairs) of a batch of n_pairs_total in local block layout.  On rank 0: *d_rows / *d_pool =
 * device memory (owned by the communicator, valid until the gather after its next, or destroy) holding the batch-global result,
 * byte for byte what one context returns for the whole batch; n_rows records of 48 bytes, n_ops CIGAR words.  On the
 * other ranks the four outputs are NULL / 0.  kslam_adopt_results_device(ctx, *d_rows, ...) makes it rank 0's result. */
kslam_status kslam_comm_gather_batch(kslam_comm *comm, uint64_t n_local_pairs, uint64_t pair_lo,
                                     uint64_t n_pairs_total, void **d_rows, uint64_t *n_rows, void **d_pool,
                                     uint64_t *n_ops);
/* The same in two halves, so that the transfer of batch k runs under the alignment of batch k + 1: _begin exchanges the
 * counts, exports this rank's records into the communicator's own buffers (the context's result is free afterwards) and
 * posts the group of sends / receives on the communicator's stream; _end waits for it and returns what _gather_batch
 * returns.  One gather in flight per communicator; rank 0 receives into two pairs of arrays in turn, so that what one
 * _end returned stays valid until the _begin after the next. */
kslam_status kslam_comm_gather_begin(kslam_comm *comm, uint64_t n_local_pairs, uint64_t pair_lo,
                                     uint64_t n_pairs_total);
kslam_status kslam_comm_gather_end(kslam_comm *comm, void **d_rows, uint64_t *n_rows, void **d_pool,
                                   uint64_t *n_ops);

/* kslam_pair_phase_a -> all-gather of the insert sizes -> kslam_pair_phase_b [-> kslam_pseudo_route -> all-to-all of the
 * heads -> kslam_pseudo_owned -> all-to-all of the scores back -> kslam_pseudo_return] on this rank's result
 * (include/kslam.h).  Afterwards the context holds this rank's read pairs / alignment pairs as after kslam_pair_screen on
 * a context that saw the whole batch.  bytes_received (may be NULL): what the exchanges brought to this rank from the
 * OTHER ranks (0 at world size 1).  KSLAM_ERR_UNSUPPORTED on every rank when the device stage declined on any
 * (an entry with more than 262144 alignment pairs): the batch's pseudo-assembly belongs to the host then. */
kslam_status kslam_comm_sharded_tail(kslam_comm *comm, int paired, uint32_t score_threshold, double score_fraction,
                                     int pseudo_assembly, kslam_pair_stats *stats, uint64_t *bytes_received);

/* What RCCL itself says about the communicator (the proof object of bench.py's `rccl` field): ncclCommCount,
 * ncclCommUserRank, ncclGetVersion, the device, and the path of the librccl the library opened. */
typedef struct {
  int32_t comm_count;
  int32_t comm_rank;
  int32_t rccl_version;
  int32_t device;
  char library[240];
} kslam_comm_facts;
kslam_status kslam_comm_info(const kslam_comm *comm, kslam_comm_facts *out);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_COMM_H_ */
