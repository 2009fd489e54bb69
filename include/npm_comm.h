/*
 * npm_comm.h -- C ABI of the data-parallel gradient exchange (RCCL over xGMI).
 *
 * The reference (levendlee/np-modeling) is single-process: it has no communication
 * code to mirror.  This is the one exchange step of the batch-sharded hot path
 * (SURVEY.md section 8e): an all-reduce of the parameter gradients between
 * Layer.backward's gradient computation and optimizer_.update (reference call sites
 * layers/mlp.py:38-39, layers/normalizations.py:73-74, layers/attentions.py:190-197).
 *
 * One process per GPU.  Rank 0 creates the id, a file of the node carries it to the other
 * ranks (host side: np_modeling_amd/parallel.py rendezvous_path), every rank calls npm_comm_init.
 * Collectives run on a dedicated communication stream so they overlap the rest of the
 * backward pass; ordering against the compute stream is by HIP events.
 */
#ifndef NPM_COMM_H
#define NPM_COMM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPM_COMM_ID_BYTES 128

enum { NPM_REDUCE_SUM = 0, NPM_REDUCE_AVG = 1, NPM_REDUCE_MAX = 2 };

const char *npm_comm_last_error(void);
/* File the RCCL entry points of this process were bound from (the shim is linked against /opt/rocm/lib/librccl.so
 * by rpath; a host program that had ALREADY loaded another librccl.so.1 -- e.g. the copy bundled with PyTorch --
 * would make the dynamic linker reuse that one: this call tells which, bench.py reports it). */
int npm_comm_library_path(char *buf, int len);
int npm_comm_unique_id(char *id /* NPM_COMM_ID_BYTES */);
/* compute_stream: npm_stream() of libnpm_hip.so (the stream gradients are produced on) */
int npm_comm_init(const char *id, int rank, int nranks, void *compute_stream);
int npm_comm_rank(int *rank, int *nranks);
/* In-place all-reduce of buf[0..count); issued on the comm stream after everything already
 * queued on the compute stream.  Returns immediately (asynchronous). */
int npm_comm_allreduce_f32(float *buf, size_t count, int op);
int npm_comm_broadcast_f32(float *buf, size_t count, int root);
/* Make the compute stream wait for every collective issued so far. */
int npm_comm_wait(void);
/* Exchange statistics, for bench.py's `exchange` object.  While enabled, every all-reduce is bracketed by timing
 * events on the communication stream and every npm_comm_wait by timing events on the compute stream;
 * npm_comm_stats synchronises both streams, returns the sums since the previous call and resets them.
 *   allreduce_ms  time the collectives occupied the communication stream (they overlap backward)
 *   exposed_ms    time the compute stream stood still in npm_comm_wait: what the exchange COSTS the step
 *   last_allreduce_ms  duration of the NEWEST all-reduce at each wait, summed: the flush a backward issues after its last
 *                 gradient exists, which no computation is left to hide (the earlier flushes overlap the rest of backward)
 * At most 8192 spans wait to be read; beyond that further ones are counted in `dropped` instead of recorded, and
 * npm_comm_stats_enable(0) returns every pending event to the pool. */
typedef struct npm_comm_exchange_stats {
    unsigned long long bytes;      /* payload bytes handed to ncclAllReduce */
    int allreduce_calls;
    int waits;
    double allreduce_ms;
    double exposed_ms;
    double last_allreduce_ms;
    int dropped;
} npm_comm_exchange_stats;
int npm_comm_stats_enable(int on);
int npm_comm_stats(npm_comm_exchange_stats *out);
/* Host-blocking: all ranks have reached this point and their GPU work is complete. */
int npm_comm_barrier(void);
/* Host scalar reduction (bench.py: max over ranks of the elapsed time). */
int npm_comm_allreduce_host_f64(double *value, int op);
int npm_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* NPM_COMM_H */
