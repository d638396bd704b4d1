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
    int32_t Tmax;
    const int32_t* rowbase; // device [Tmax]
    const int32_t* T_b;     // device [B] (sorted, descending)
    float max_act;          // <= 0: no ceiling
    // Exchange buffer: chunk-major copy of the state, [2 groups][Hp/16][n_xrows][16] floats.
    // Exchange row of utterance b at STEP j = xbase[j] + b; xbase rounds every step's
    // block up to 4 rows (256 B) so that no cache line holds data of two different steps,
    // and up to 16 rows from 17 alive utterances on (recurrent_step_xrows below).
    float* xbuf;
    const int32_t* xbase;   // device [Tmax]
    int32_t n_xrows;
    unsigned* counters;     // [REC_COUNTER_WORDS]: sticky error word at [2], per-producer step flags from [32]
    int32_t sync_mode;      // 0: plain exchange stores + agent-scope release fence before the flag
                            // 1: write-through (sc1) exchange stores, no fence
    int32_t poll_delay;     // s_sleep units (64 cycles) before the first poll of a step; < 0: pick by layer size
    int32_t variant;        // 0: pick automatically; 1: force the one-workgroup-per-CU kernel;
                            // 2: two-chain kernel with the linear (not XCD-grouped) block map;
                            // 3: force the non-persistent per-step fallback;
                            // 4: two workgroups per CU for 17..32 utterances (round 1-3 kernel);
                            // 8..23: both chains in one 8-wave workgroup, mode = variant - 8;
                            // 40: more than 32 utterances without the load / MFMA pipelining of round 5;
                            // 42: 6..16 utterances (fp32) on the sentinel / MFMA kernel instead of the single-chain flag kernel;
                            // 45: 33..48 utterances as ONE launch of the one-workgroup-per-CU kernel (default: 32 + rest);
                            // 43: the sentinel / VALU kernel for up to 8 utterances (default: up to 3); 44: the single-chain
                            //     flag kernel from 1 utterance (default: from 4)
                            // 46: exchange tiles row-major throughout (rounds 1-5a; bit-identical A/B of the lane-order layout);
                            // 47: more than 32 utterances on the one-slab-per-CU kernel of rounds 1-5 (default since round 6 at 1824 / 2048
                            //     units: units x utterances, brnn_recurrent_t_kernel); 50: the tiled kernel at 512 / 1024 units as well
                            //     (no faster there; tests); 49: DIAGNOSTIC, wrong results -- the tiled kernel re-reads step 0's exchange rows
    unsigned* debug;        // nullable: s_memtime stamps [2 wgs][16 steps][8] for steps 64..79
    int32_t b_off;          // rank of this launch's first utterance in the packed minibatch (minibatches of
                            // more than 128 utterances run as several launches; T_b already points at it)
    const int32_t* T_host;  // nullable: host copy of the full (sorted) T_b, used to shorten later launches
    int32_t linear_map;     // two-chain kernel: linear block -> (chain, producer) map, chain = direction (the single-chain launch of
                            // 4..16 utterances; set by the launcher -- `variant` keeps its A/B meaning, e.g. 46, there too)
    int32_t prec16;         // != 0: "fp16 activations" -- the 6..16-utterance kernel exchanges the state
                            // and holds the weights in 16 bit (float16 forward, bfloat16 BPTT), fp32 accumulate
};
// [2 wgs][16 steps][8] s_memtime stamps of steps 64..79, then [512 workgroups][8] wall-clock
// (100 MHz, global) stamps of step 70 for every workgroup
static constexpr int REC_DEBUG_ALL_OFF = 2 * 16 * 8;
static constexpr int REC_DEBUG_WORDS = 2 * 16 * 8 + 512 * 8;
#ifndef SCTC_REC_FLAG_STRIDE
#define SCTC_REC_FLAG_STRIDE 8
#endif
// words between the step flags of two producers (32 = one 128-byte line per flag)
static constexpr int REC_FLAG_STRIDE = SCTC_REC_FLAG_STRIDE;
static constexpr int REC_COUNTER_WORDS = 32 + 4 * 128 * REC_FLAG_STRIDE;   // error word + up to 4 chains x 128 producers

// Exchange rows of a time step whose `alive` utterances are still running: a multiple of 4 (256 bytes: no cache line
// holds data of two steps); from 17 utterances on a multiple of 16, so that every tile of 16 utterances that is alive at
// all has its 16 rows inside the step's block and keeps the lane-order layout (recurrent.hip, lane_order_steps).
static inline int recurrent_step_xrows(int alive) { return alive >= 17 ? (alive + 15) & ~15 : (alive + 3) & ~3; }
// exchange rows needed for `rows` frames spread over `tmax` time steps (worst case: 3 rows of padding per step, 12 more
// for each of the at most rows / 17 steps with 17 or more utterances alive)
static inline int64_t recurrent_xrows_bound(int64_t rows, int64_t tmax) { return rows + 3 * tmax + 12 * (rows / 17 + 1); }
size_t recurrent_xbuf_floats(int Hp, int64_t max_xrows);
int recurrent_supported(int Hp, int B, char* why, int why_len);
// which path a launch_recurrent call took (`path` out-parameter, nullable)
static constexpr int REC_PATH_NONE = 0;
static constexpr int REC_PATH_PERSISTENT = 1;          // one whole-device persistent launch per pass
static constexpr int REC_PATH_PERSISTENT_LEASED = 2;   // the same under the inter-process device lease
static constexpr int REC_PATH_FALLBACK = 3;            // one launch per time step (no spinning)
int launch_recurrent(const RecArgs& a, hipStream_t stream, int* path = nullptr);
// clears the sticky error word (counters[2]): once per step, before its first recurrent launch
int recurrent_clear_error(unsigned* counters, hipStream_t stream);
// shared-device mode: persistent launches take an inter-process lease (flock on a per-device file)
// and run synchronously.  Initial value: SCTC_SHARED_DEVICE in the environment.
int recurrent_shared_device_mode();
void recurrent_set_shared_device_mode(int on);

}  // namespace sctc
