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

// ReplayBuffer.add's device side (replay_buffer.py:185-196): everything the host accumulator changed since the last flush --
// new frames, rewritten element rows, moved sampler indices -- arrives in ONE staged buffer and is scattered by one launch.
// Workgroups [0, n_frames): one frame each (16-byte pieces); then the element rows; then the index -> slot pairs.
__global__ __launch_bounds__(256) void apply_staged_kernel(const uint8_t* __restrict__ st, const isdqn_staged_updates u, int row_blocks,
                                                           uint8_t* __restrict__ frames, int64_t frame_stride,
                                                           int* __restrict__ elem_frames, int* __restrict__ elem_action,
                                                           float* __restrict__ elem_reward, uint8_t* __restrict__ elem_terminal,
                                                           int* __restrict__ index_to_slot) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b < u.n_frames) {
        const int slot = reinterpret_cast<const int*>(st + u.off_frame_slots)[b];
        const uint8_t* src = st + u.off_frame_data + (int64_t)b * u.frame_bytes;
        uint8_t* dst = frames + (int64_t)slot * frame_stride;
        // 16-byte pieces only when both sides are 16-byte aligned for EVERY frame: frame size and stride multiples of 16 and
        // the bases of the staged frame block and of the frame store aligned (the host packer rounds its sections to 16 bytes,
        // but this is a public entry point); otherwise bytes
        const bool wide = ((u.frame_bytes | (int)(frame_stride & 15)) & 15) == 0 &&
                          ((reinterpret_cast<uintptr_t>(st + u.off_frame_data) | reinterpret_cast<uintptr_t>(frames)) & 15) == 0;
        const int n16 = wide ? u.frame_bytes / 16 : 0;
        for (int i = t; i < n16; i += 256) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (int i = n16 * 16 + t; i < u.frame_bytes; i += 256) dst[i] = src[i];
        return;
    }
    const int rb = b - u.n_frames;
    if (rb < row_blocks) {
        const int i = rb * 256 + t;
        if (i < u.n_rows * u.stack2) {
            const int r = i / u.stack2, c = i - r * u.stack2;
            const int row = reinterpret_cast<const int*>(st + u.off_rows)[r];
            elem_frames[(int64_t)row * u.stack2 + c] = reinterpret_cast<const int*>(st + u.off_row_frames)[i];
            if (c == 0) {
                elem_action[row] = reinterpret_cast<const int*>(st + u.off_row_action)[r];
                elem_reward[row] = reinterpret_cast<const float*>(st + u.off_row_reward)[r];
                elem_terminal[row] = (st + u.off_row_terminal)[r];
            }
        }
        return;
    }
    const int i = (rb - row_blocks) * 256 + t;
    if (i < u.n_index) index_to_slot[reinterpret_cast<const int*>(st + u.off_index_rows)[i]] = reinterpret_cast<const int*>(st + u.off_index_vals)[i];
}

extern "C" int isdqn_replay_apply_staged(const uint8_t* staged, const isdqn_staged_updates* u, uint8_t* frames, int64_t frame_stride,
                                         int32_t* elem_frames, int32_t* elem_action, float* elem_reward, uint8_t* elem_terminal,
                                         int32_t* index_to_slot, void* stream) {
    ISDQN_REQUIRE(staged && u, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(u->n_frames >= 0 && u->n_rows >= 0 && u->n_index >= 0 && u->stack2 >= 1 && u->frame_bytes >= 0, ISDQN_ERR_ARG, "negative count");
    ISDQN_REQUIRE(u->n_frames == 0 || frames, ISDQN_ERR_ARG, "null frame store");
    ISDQN_REQUIRE(u->n_rows == 0 || (elem_frames && elem_action && elem_reward && elem_terminal), ISDQN_ERR_ARG, "null element table");
    ISDQN_REQUIRE(u->n_index == 0 || index_to_slot, ISDQN_ERR_ARG, "null index table");
    // the int32 / float sections are read as such: 4-byte aligned offsets on a 4-byte aligned base
    ISDQN_REQUIRE((reinterpret_cast<uintptr_t>(staged) & 3) == 0 &&
                  ((u->off_frame_slots | u->off_rows | u->off_row_frames | u->off_row_action | u->off_row_reward | u->off_index_rows |
                    u->off_index_vals) & 3) == 0,
                  ISDQN_ERR_ARG, "staged int32/float sections must be 4-byte aligned");
    ISDQN_REQUIRE(u->off_frame_slots >= 0 && u->off_frame_data >= 0 && u->off_rows >= 0 && u->off_row_frames >= 0 && u->off_row_action >= 0 &&
                  u->off_row_reward >= 0 && u->off_row_terminal >= 0 && u->off_index_rows >= 0 && u->off_index_vals >= 0,
                  ISDQN_ERR_ARG, "negative section offset");
    ISDQN_REQUIRE(u->n_frames == 0 || frame_stride >= u->frame_bytes, ISDQN_ERR_SHAPE, "frame_stride < frame_bytes");
    const int row_blocks = (u->n_rows * u->stack2 + 255) / 256, idx_blocks = (u->n_index + 255) / 256;
    const int blocks = u->n_frames + row_blocks + idx_blocks;
    if (blocks == 0) return ISDQN_OK;
    hipLaunchKernelGGL(apply_staged_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, staged, *u, row_blocks, frames, frame_stride,
                       elem_frames, elem_action, elem_reward, elem_terminal, index_to_slot);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

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
