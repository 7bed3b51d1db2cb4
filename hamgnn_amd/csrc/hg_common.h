// hg_common.h -- shared helpers of libhamgnn_hip.so (error reporting; no torch, no CUDA-compat shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int hg_fail(int code, const char* msg);          // records msg for hg_last_error(), returns code
int hg_check_launch(const char* what);           // hipGetLastError() after a launch -> 0 / negative

struct HgWigOff { int o[8]; };                   // float offsets of D^l inside one packed Wigner row (host array -> kernarg)

// The stream decides the device: an entry point runs its launches (and its one-time kernel-attribute calls) with the stream's device
// current, so one host thread can drive several GPUs; the previous device is restored on exit.  The NULL stream means "current device".
struct HgDeviceGuard {
    int prev = 0, dev = 0;
    bool switched = false;
    explicit HgDeviceGuard(void* stream) {
        (void)hipGetDevice(&prev);
        dev = prev;
        hipDevice_t d;
        if (stream && hipStreamGetDevice((hipStream_t)stream, &d) == hipSuccess) dev = (int)d;
        if (dev != prev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~HgDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: done once per (kernel, device); thread-safe
// (the call is idempotent, so a race only repeats it).  Not a stream operation, hence legal outside of -- but not during -- graph
// capture: warm every device up once before capturing.
#define HG_MAX_DEVICES 64
int hg_lds_attr_once(unsigned char* done_flags /*[HG_MAX_DEVICES]*/, int dev, const void* kernel, int bytes);
