// hg_common.h -- shared helpers of libhamgnn_hip.so (error reporting; no torch, no CUDA-compat shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int hg_fail(int code, const char* msg);          // records msg for hg_last_error(), returns code
int hg_check_launch(const char* what);           // hipGetLastError() after a launch -> 0 / negative

struct HgWigOff { int o[8]; };                   // float offsets of D^l inside one packed Wigner row (host array -> kernarg)
