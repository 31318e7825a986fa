// hip_errors.h -- what happens when the GPU cannot serve a call (round 5).
//
// The reference's MSM symbols return void and have no error channel (bindings/c_curve_decls_parallel.nim:26-28); its protocol
// symbols return a status enum and never terminate the process.  This library has no CPU path to fall back to, so:
//   * every entry point that HAS a return value reports a GPU refusal through it and records WHY in a per-thread "last error"
//     (ctt_hip_last_error / ctt_hip_last_error_message): -1 refused (bad arguments), -5 busy (all in-flight slots taken), -2 out of device
//     memory, -3 no usable HIP device, -4 a HIP runtime call failed (the context it happened on is marked lost and refuses
//     every later call);
//   * the Constantine-named `void` MSM symbols, which cannot report anything, abort with the same message.
// Rounds 1-4 aborted on every failed HIP call, inside the protocol symbols too (an Ethereum client linking the
// library could be taken down by GPU memory pressure while it processes attacker-supplied precompile input).
//
// Mechanism: HIP_CHECK calls hip_failed(), which records the error and THROWS HipFailure when the calling thread is inside an
// ErrorGuard (an entry point with an error channel) and aborts otherwise (a void symbol; a helper thread, whose exception
// nobody could catch).
#pragma once
#include <stddef.h>

namespace ctt {

// (ERR_BUSY, round 6: "all in-flight slots taken" used to share -1 with bad arguments, and the EVM symbols' retry loop could not tell them apart)
enum { ERR_NONE = 0, ERR_REFUSED = -1, ERR_OUT_OF_MEMORY = -2, ERR_NO_DEVICE = -3, ERR_HIP_FAILURE = -4, ERR_BUSY = -5 };

struct HipFailure { int code; };

struct ErrorState {
  int code = 0;
  int guard = 0;       // > 0: inside an entry point that can report
  char msg[320] = {0};
};
ErrorState& error_state();                                    // per thread (msm_engine.hip)
void set_last_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// a failed HIP runtime call: last error = ERR_HIP_FAILURE (ERR_OUT_OF_MEMORY for an allocation), then throw or abort (see above)
[[noreturn]] void hip_failed(const char* expr, const char* what, int hip_error_is_oom, const char* file, int line);

// The calling thread's current HIP device is the caller's business (a host program with HIP code of its own, torch): an entry point
// selects its context's device for the duration of the call and puts the previous one back (round 5; rounds 1-4 left the context's
// device selected -- after ctt_hip_msm_set_devices({0..7}) a torch process found itself on GPU 7).
struct DeviceScope {
  int prev = -1, dev;
  explicit DeviceScope(int device);
  ~DeviceScope();
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

struct ErrorGuard {
  ErrorGuard() { error_state().guard++; }
  ~ErrorGuard() { error_state().guard--; }
  ErrorGuard(const ErrorGuard&) = delete;
  ErrorGuard& operator=(const ErrorGuard&) = delete;
};

}  // namespace ctt
