// Serial chains: several small dependent launches of a plan recorded and run as ONE run-time generated kernel (one
// workgroup of 1024 threads executing them back to back; csrc/jit.hip, bodies in chain_bodies.cuh).
//
//   hb_chain_begin();                       the calling thread starts recording
//   ... chain-aware entry points ...        a SMALL launch they would issue is recorded instead (hb_chain_push); before any
//                                           launch they cannot record they run what is recorded (hb_chain_flush), so the
//                                           order of the calls is the order of execution
//   hb_chain_end(stream);                   runs what is still recorded, stops recording
//
// Only entry points that know about chains may be called between begin and end (henbun_amd/graph.py wraps exactly those
// steps): hb_ewise_jit_run, hb_gauss_ll, hb_adam_step, hb_sgp_fwd (its finishing pass), hb_gram_bwd (its lengthscale fold).
// Where hiprtc is unavailable hb_chain_begin records nothing and every call launches as usual.
#ifndef HB_CHAIN_CUH
#define HB_CHAIN_CUH
#include "common.cuh"
#include "ew_prog.cuh"

enum { HB_CHAIN_PROG = 1, HB_CHAIN_ADAM = 2, HB_CHAIN_GLL = 3, HB_CHAIN_SGP_FINISH = 4, HB_CHAIN_GRAM_ELL = 5, HB_CHAIN_GLL_FOLD = 6 };
#define HB_CHAIN_MAX_JOBS 6
// A chain is ONE workgroup, and one CU streams ~10 bytes per cycle (24 GB/s): a job that moves more than a few tens
// of KB costs more inside a chain than the ~4.5 us kernel boundary it saves (measured: the sparse-GP finishing pass at
// cfg 2 -- 0.5 MB of column partials -- took 12 us in a chain against 4.7 us as 16 workgroups).  Size limits per kind:
#define HB_CHAIN_PROG_MAX_N 4096      // elements of an elementwise program
#define HB_CHAIN_FINISH_MAX_N 512     // columns (x experts) of a sparse-GP finishing pass
#define HB_CHAIN_GLL_MAX_N 4096       // points of a likelihood head
#define HB_CHAIN_ADAM_MAX_N 4096      // parameters of an Adam update
#define HB_CHAIN_ELL_MAX_N 16384      // partials of a lengthscale fold
#define HB_CHAIN_GLL_FOLD_MAX_N 4096  // units of a likelihood-head fold

struct HbChainJob {
  int kind = 0, is64 = 0;
  const void* p[HB_PROG_MAX_IN + HB_PROG_MAX_OUT] = {};   // pointers (programs: the nin inputs, then the nout outputs)
  long l[8] = {};
  double d[8] = {};
  ProgArgs prog;      // HB_CHAIN_PROG: the validated descriptor
  int reduces = 0;
};

bool hb_chain_recording();
int hb_chain_push(const HbChainJob& job, hipStream_t stream);
int hb_chain_flush(hipStream_t stream);

#endif  // HB_CHAIN_CUH
