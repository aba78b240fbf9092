// sqg_hip.hip -- MI355X (gfx950) implementation of include/sqg.h.
//
// One translation unit: the gfx950 kernels (sqg_kernels.h: k_common.h, k_events.h, k_samples.h, k_sampler.h,
// k_svb.h) and the host side of the C ABI (h_*.h: context/batch management, staging, launches, results).
// No MFMA anywhere: this is an integer-LCG / transcendental / streaming-store path.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see squigulator_amd/build.py).
//
// The product path never touches oracle/: this file is self-contained.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/sqg.h"

#include "sqg_kernels.h"

// the host side, in dependency order
#include "h_common.h"     // sqg_ctx, sqg_batch, HIPCHK, host_nrng, prefix constants
#include "h_context.h"    // sqg_create, sqg_destroy, ...
#include "h_stage.h"      // sqg_batch_stage, sqg_batch_free
#include "h_sampler.h"    // sqg_genome_load, sqg_batch_sample, sqg_batch_sample_range, sqg_skip_reads, sqg_fetch_reads
#include "h_run.h"        // sqg_batch_run, sqg_batch_run_begin / _end, sqg_set_range_mode
#include "h_results.h"    // sqg_batch_wait, sqg_fetch_*, sqg_batch_compress, sqg_get_timing, ...
#include "h_blow5.h"      // sqg_blow5_*: the native BLOW5 writer
