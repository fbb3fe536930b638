// lib.hip - library-level entry points of libsslam_hip.so (version, launch counter).
#include "common.h"

long long g_sslam_launches = 0;

extern "C" int sslam_version(void) { return 100; }
extern "C" const char *sslam_arch(void) { return "gfx950"; }
extern "C" long long sslam_launch_count(void) { return g_sslam_launches; }
