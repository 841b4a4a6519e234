// Device-resident replay: byte/integer HBM-bound kernels for ReplayBuffer.sample
// (slimdqn/sample_collection/replay_buffer.py:198-213).
//
// HBM layout: frames[slot][h*w] uint8 (one single frame per env step, 7,056 B for Atari)
// and an element table in SoA form indexed by element slot.  A sampled batch is B rows of
// that table; the pixels themselves are never copied for the training step (the first
// convolution reads the frames through the id table).  `materialize` exists for callers
// that want the reference's (B, h, w, stack) arrays.
#include "common.h"

namespace isdqn {

__global__ __launch_bounds__(256) void gather_rows_kernel(const int* __restrict__ elem_frames,
                                                          const int* __restrict__ elem_action,
                                                          const float* __restrict__ elem_reward,
                                                          const uint8_t* __restrict__ elem_terminal, int stack2,
                                                          const int* __restrict__ index_to_slot,
                                                          const int* __restrict__ slots, int B,
                                                          int* __restrict__ out_ids, int* __restrict__ out_action,
                                                          float* __restrict__ out_reward,
                                                          uint8_t* __restrict__ out_terminal) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * stack2) return;
    int b = i / stack2, c = i - b * stack2;
    int slot = slots[b];
    if (index_to_slot != nullptr) slot = index_to_slot[slot];
    out_ids[i] = elem_frames[(int64_t)slot * stack2 + c];
    if (c == 0) {
        out_action[b] = elem_action[slot];
        out_reward[b] = elem_reward[slot];
        out_terminal[b] = elem_terminal[slot];
    }
}

// One workgroup per (sample, which): interleave `stack` planar frames into (h, w, stack).
// Reads: coalesced 4-byte words of each plane; writes: 4*stack bytes per thread, contiguous.
__global__ __launch_bounds__(256) void materialize_kernel(const uint8_t* __restrict__ frames, int64_t frame_stride,
                                                          int hw, int stack, const int* __restrict__ frame_ids,
                                                          uint8_t* __restrict__ out_state,
                                                          uint8_t* __restrict__ out_next) {
    const int b = blockIdx.x, which = blockIdx.y;
    const int* ids = frame_ids + ((int64_t)b * 2 + which) * stack;
    uint8_t* out = (which ? out_next : out_state) + (int64_t)b * hw * stack;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        for (int c = 0; c < stack; ++c) {
            int id = ids[c];
            out[(int64_t)p * stack + c] = id < 0 ? (uint8_t)0 : frames[(int64_t)id * frame_stride + p];
        }
    }
}

__global__ __launch_bounds__(256) void deinterleave_kernel(const uint8_t* __restrict__ state,
                                                           const uint8_t* __restrict__ next_state, int hw, int stack,
                                                           uint8_t* __restrict__ out_frames,
                                                           int* __restrict__ out_ids) {
    const int b = blockIdx.x, which = blockIdx.y;
    const uint8_t* in = (which ? next_state : state) + (int64_t)b * hw * stack;
    const int64_t first = ((int64_t)b * 2 + which) * stack;
    for (int p = threadIdx.x; p < hw; p += blockDim.x)
        for (int c = 0; c < stack; ++c) out_frames[(first + c) * hw + p] = in[(int64_t)p * stack + c];
    if (threadIdx.x < stack) out_ids[first + threadIdx.x] = (int)(first + threadIdx.x);
}

}  // namespace isdqn

using namespace isdqn;

extern "C" int isdqn_replay_gather_rows(const int32_t* elem_frames, const int32_t* elem_action,
                                        const float* elem_reward, const uint8_t* elem_terminal, int32_t stack,
                                        const int32_t* index_to_slot, const int32_t* slots, int32_t B,
                                        int32_t* out_frame_ids, int32_t* out_action,
                                        float* out_reward, uint8_t* out_terminal, void* stream) {
    ISDQN_REQUIRE(elem_frames && elem_action && elem_reward && elem_terminal && slots, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(out_frame_ids && out_action && out_reward && out_terminal, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(stack >= 1 && B >= 0, ISDQN_ERR_SHAPE, "bad shape");
    if (B == 0) return ISDQN_OK;
    int total = B * 2 * stack;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, elem_frames,
                       elem_action, elem_reward, elem_terminal, 2 * stack, index_to_slot, slots, B, out_frame_ids, out_action,
                       out_reward, out_terminal);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

extern "C" int isdqn_replay_materialize(const uint8_t* frames, int64_t frame_stride, int32_t h, int32_t w,
                                        int32_t stack, const int32_t* frame_ids, int32_t B, uint8_t* out_state,
                                        uint8_t* out_next_state, void* stream) {
    ISDQN_REQUIRE(frames && frame_ids && out_state && out_next_state, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(h > 0 && w > 0 && stack >= 1 && B >= 0 && frame_stride >= (int64_t)h * w, ISDQN_ERR_SHAPE,
                  "bad shape");
    if (B == 0) return ISDQN_OK;
    hipLaunchKernelGGL(materialize_kernel, dim3(B, 2), dim3(256), 0, (hipStream_t)stream, frames, frame_stride, h * w,
                       stack, frame_ids, out_state, out_next_state);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

extern "C" int isdqn_replay_deinterleave(const uint8_t* state, const uint8_t* next_state, int32_t h, int32_t w,
                                         int32_t stack, int32_t B, uint8_t* out_frames, int32_t* out_frame_ids,
                                         void* stream) {
    ISDQN_REQUIRE(state && next_state && out_frames && out_frame_ids, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(h > 0 && w > 0 && stack >= 1 && stack <= 256 && B >= 0, ISDQN_ERR_SHAPE, "bad shape");
    if (B == 0) return ISDQN_OK;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(B, 2), dim3(256), 0, (hipStream_t)stream, state, next_state, h * w,
                       stack, out_frames, out_frame_ids);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}
