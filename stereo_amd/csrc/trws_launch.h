// Host-side launchers of the TRW-S sweep kernel families (one translation unit each).
// what: 0 = forward sweep, 1 = backward sweep, 2 = forward sweep + primal pass of the previous
// iteration, 3 = primal pass only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "trws_dev.h"

namespace stereo {

size_t generic_lds_bytes(int Kp);
void generic_set_attributes(int lds);
void launch_generic(int kernel, int mode, int what, int blocks, size_t lds, hipStream_t s, const DevParams &p, int epoch);

size_t pipe_lds_bytes();
int pipe_threads();
void pipe_set_attributes();
void launch_pipe(int kernel, bool shared, int what, int blocks, hipStream_t s, const DevParams &p, int epoch);
void launch_pipe_group(int kernel, bool shared, int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch);

size_t pipe2_lds_bytes();
void pipe2_set_attributes();
void launch_pipe2(int kernel, bool shared, int what, int blocks, hipStream_t s, const DevParams &p, int epoch);
void launch_pipe2_group(int kernel, bool shared, int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch);

size_t wide_lds_bytes();
void wide_set_attributes();
void launch_wide(int kernel, int what, int blocks, hipStream_t s, const DevParams &p, int epoch);
void launch_wide_group(int kernel, int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch);


}  // namespace stereo
