// Persistent weight-stationary kernel for the bi-directional recurrence
// (brnnet.py:143-153 forward, :206-224 backprop through time).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sctc {

// Two independent recurrences ("groups") run in one launch on disjoint CUs:
//   forward pass : group 0 = hFor (time ascending, Wf), group 1 = hBack (descending, Wb)
//   BPTT         : group 0 = deltasFor (descending, Wf^T, mask from hFor),
//                  group 1 = deltasBack (ascending, Wb^T, mask from hBack)
// Rows follow the packed time-major minibatch layout: frame t of utterance b
// (utterances sorted by length, longest first) is row rowbase[t] + b.
struct RecArgs {
    const float* W[2];      // [Hp][ldw] row-major recurrent weight of each group
    int64_t ldw;
    int32_t transpose;      // 0: out = W x (forward); 1: out = W^T x (BPTT)
    int32_t descending[2];  // time order of each group
    const float* pre[2];    // [rows][ld]: z (forward) / deltasOut (BPTT): the term added per frame
    const float* act[2];    // BPTT only: stored hFor/hBack for the (0,maxAct) mask; else NULL
    float* out[2];          // [rows][ld]
    int64_t ld;
    int32_t Hp;             // padded layer size (multiple of 32)
    int32_t B;              // utterances
    int32_t Bp;             // B rounded up to 16
    int32_t Tmax;
    const int32_t* rowbase; // device [Tmax]
    const int32_t* nact;    // device [Tmax] utterances still running at step j
    const int32_t* T_b;     // device [B] (sorted, descending)
    float max_act;          // <= 0: no ceiling
    float* xbuf;            // exchange buffers [2 groups][2 parity][Hp/16][Bp][16]
    unsigned* counters;     // [2] arrival counters (zeroed by the launcher) + [1] error word
    int32_t sync_mode;      // 0: plain stores/loads + agent release/acquire fences
                            // 1: write-through (sc1) 8-byte atomics both sides, no fences
};

size_t recurrent_xbuf_floats(int Hp, int B);
int recurrent_supported(int Hp, int B, char* why, int why_len);
int launch_recurrent(const RecArgs& a, hipStream_t stream);

}  // namespace sctc
