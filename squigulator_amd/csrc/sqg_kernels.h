// sqg_kernels.h -- gfx950 device code of the per-read signal path (included by sqg_hip.hip); the kernels live in
// k_common.h, k_events.h, k_part.h, k_part_events.h, k_samples.h, k_sampler.h, k_svb.h and k_blow5.h.
//
//   k_init_rows   per-(worker,k-mer) stream seeds                       (src/sim.c:238-257)
//   k_dwell       per-event dwell draw from the worker's time stream   (src/gensig.c:254-257)
//   k_scan        read lengths -> output offsets
//   k_events      per worker chain: ranks, in-order hand-out of the k-mer streams
//   k_part_*      k > 6, few workers: the stream hand-out over events bucketed by the top bits of the rank (k_part.h)
//   k_samples     per 64-event tile: the samples                       (src/gensig.c:226-356)
//   k_fixup       FP64 recomputation of the samples the certified fp32 path could not decide
//   k_certify     exhaustive error sweep of the fp32 normal-deviate path over all 2^31-2 states
//   k_store_probe int16 streaming-store ceiling
//   k_items       one descriptor per work item of the lean sample kernel
//   k_sample, k_copy_reads   gen_read on the device-resident genome        (src/genread.c:125-370)
//   k_svb_*       slow5lib's svb-zd signal compression                    (slow5lib/src/slow5_press.c:1055-1087)
//   k_blow5_frame BLOW5 records (slow5_rec_to_mem's layout) in stored-block zlib streams (slow5lib/src/slow5.c:3928-4072)
//
// Arithmetic modes.  EXACT: every draw goes through the FP64 restatement of nrng()
// (src/rand.h:87-94).  CERTIFIED: a draw is first evaluated with fp32 hardware transcendentals;
// the result is accepted only if the digitised value provably cannot differ from the FP64 one
// (|frac - 1/2| test against a bound built from the swept error delta_x), and
// is otherwise recomputed in FP64.  Both modes produce identical int16 streams.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>


#include "k_common.h"
#include "k_events.h"
#include "k_part.h"
#include "k_part_events.h"
#include "k_samples.h"
#include "k_sampler.h"
#include "k_svb.h"
#include "k_blow5.h"
