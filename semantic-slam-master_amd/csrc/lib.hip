// lib.hip - library-level entry points of libsslam_hip.so (version, launch counter, test-only knobs).
#include <stdlib.h>
#include <string.h>

#include "common.h"

long long g_sslam_launches = 0;
long long g_sslam_knob[KNOB_COUNT];

namespace {
long long g_knob_at_load[KNOB_COUNT];   // what the environment said when the library was loaded: "unset" restores THIS
const char *const kKnobNames[KNOB_COUNT] = {
    "SSLAM_M1_VARIANT",   "SSLAM_CONV_VARIANT",  "SSLAM_CONV_LATENCY_ROWS", "SSLAM_CONV_LAT2_ROWS",   "SSLAM_CONV_NO_HALO",
    "SSLAM_CONV_TAIL",    "SSLAM_CONVBF_NO_HALO", "SSLAM_CONVBF_TAIL",      "SSLAM_CONVBF_VARIANT",   "SSLAM_VIT_NO_FUSED_MLP",
    "SSLAM_BN_FORM",      "SSLAM_VIT_F32_NO_KEY_SPLIT", "SSLAM_RT_STOP"};
// the ONE place the environment is read: when the library is loaded
struct KnobInit {
    KnobInit() {
        for (int i = 0; i < KNOB_COUNT; i++) {
            const char *e = getenv(kKnobNames[i]);
            g_sslam_knob[i] = g_knob_at_load[i] = e ? atoll(e) : SSLAM_KNOB_UNSET;
        }
    }
} g_knob_init;
}  // namespace

extern "C" int sslam_version(void) { return 300; }
extern "C" const char *sslam_arch(void) { return "gfx950"; }
extern "C" long long sslam_launch_count(void) { return __atomic_load_n(&g_sslam_launches, __ATOMIC_RELAXED); }

// One caller-owned scratch buffer that serves every *_ws entry of a pipeline step enqueued on ONE stream (the stages run
// in stream order, so they can share it): the larger of the per-entry needs.
extern "C" long long sslam_workspace_bytes(int n_frames, int G, int K, int n_pairs) {
    if (n_frames <= 0 || G <= 0 || K <= 0 || n_pairs < 0) return SSLAM_E_INVALID;
    const long long a = sslam_selector_saliency_workspace_bytes(n_frames, G);
    const long long b = n_pairs > 0 ? sslam_sim_argmax_workspace_bytes(K, n_pairs) : 0;
    const long long m = a > b ? a : b;
    return (m + 255) & ~255LL;
}

extern "C" int sslam_test_set_knob(const char *name, long long value, int unset) {
    if (!name) return SSLAM_E_INVALID;
    for (int i = 0; i < KNOB_COUNT; i++)
        if (!strcmp(name, kKnobNames[i])) {
            // unset: back to the load-time value (an SSLAM_* variable exported for a whole test session survives the tests
            // that flip the same knob), not to the built-in default
            g_sslam_knob[i] = unset ? g_knob_at_load[i] : value;
            return SSLAM_OK;
        }
    return SSLAM_E_INVALID;
}
