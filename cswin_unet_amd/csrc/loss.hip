// Segmentation loss of the training step: 0.4 * CrossEntropy + 0.6 * soft-Dice (trainer.py:55-57, utils.py:9-45),
// fused into one pass over the (B, ncls, H, W) logits each way: no one-hot tensor, no softmax tensor,
// no per-class .item() host syncs.  The Dice ratio is formed from batch-GLOBAL sums
// (intersect, y_sum, z_sum per class): `cswin_loss_sums` leaves the 1 + 3*ncls partial sums in device
// memory so a data-parallel job can all-reduce exactly those floats before `cswin_loss_finalize`
// (the reference's DataParallel computes the loss on the gathered global batch).
#include "common.h"

namespace {

// sums layout: [0] = sum over pixels of -log p[label];  [1 + c] = intersect_c;  [1 + ncls + c] = y_sum_c;  [1 + 2 ncls + c] = z_sum_c
// PROBS: the input already holds class probabilities (DiceLoss(..., softmax=False), utils.py:32-34): no softmax here
template <int NC, bool PROBS>
__global__ __launch_bounds__(256) void loss_sums_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                         float* __restrict__ partial, int B, long HW) {
    constexpr int NV = 1 + 3 * NC;
    __shared__ float red[4][NV];
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / HW, p = i - b * HW;
        const float* lp = logits + b * NC * HW + p;
        float v[NC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            v[c] = lp[c * HW];
            mx = fmaxf(mx, v[c]);
        }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!PROBS) v[c] = __expf(v[c] - mx);
            sum += v[c];
        }
        const float inv = PROBS ? 1.0f : 1.0f / sum;
        const int lab = (int)labels[i];
        // nn.CrossEntropyLoss raises on a target outside [0, ncls) (trainer.py:40,55); without a host sync the device-side
        // equivalent is to poison the CE sum: the step's loss reads NaN instead of a silently biased value
        if ((unsigned)lab >= (unsigned)NC) acc[0] = __builtin_nanf("");
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float pc = v[c] * inv;
            const float oh = (c == lab) ? 1.f : 0.f;
            acc[0] -= oh * __logf(fmaxf(pc, 1e-37f));
            acc[1 + c] += pc * oh;
            acc[1 + NC + c] += oh;
            acc[1 + 2 * NC + c] += pc * pc;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV)
        partial[(long)blockIdx.x * NV + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0..2] = loss, ce, dice;  coef[c] = a_c, coef[ncls + c] = b_c with d dice_c / d p_c(pixel) = a_c * onehot + b_c * p
__global__ void loss_finalize_kernel(const float* __restrict__ sums, float* __restrict__ out, float* __restrict__ coef,
                                     float n_pixels, int ncls, float w_ce, float w_dice, const float* __restrict__ class_weight) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float smooth = 1e-5f;
    const float ce = sums[0] / n_pixels;
    float dice = 0.f;
    for (int c = 0; c < ncls; ++c) {
        const float I = sums[1 + c], Y = sums[1 + ncls + c], Z = sums[1 + 2 * ncls + c];
        const float D = Z + Y + smooth;
        const float wc = class_weight ? class_weight[c] : 1.f;          // utils.py:44: loss += dice * weight[i]
        dice += wc * (1.f - (2.f * I + smooth) / D);
        coef[c] = wc * -2.f / D;
        coef[ncls + c] = wc * 2.f * (2.f * I + smooth) / (D * D);
    }
    dice /= ncls;
    // an out-of-range label poisons the CE sum (NaN); utils.DiceLoss (w_ce = 0) one-hots with == and simply ignores such a label
    // (utils.py:13-19), so the CE term must not reach a loss that has none
    out[0] = (w_ce != 0.f ? w_ce * ce : 0.f) + w_dice * dice;
    out[1] = ce;
    out[2] = dice;
}

// dlogits = gout * [ ce_scale * (p - onehot) + dice_scale * p_c * (G_c - sum_j p_j G_j) ],  G_c = a_c onehot_c + b_c p_c
template <int NC, bool PROBS>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                        const float* __restrict__ coef, const float* __restrict__ gout,
                                                        float* __restrict__ dlogits, float ce_scale, float dice_scale,
                                                        int B, long HW) {
    float a[NC], bb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        a[c] = coef[c];
        bb[c] = coef[NC + c];
    }
    const float g = gout ? gout[0] : 1.f;
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / HW, p = i - b * HW;
        const float* lp = logits + b * NC * HW + p;
        float v[NC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            v[c] = lp[c * HW];
            mx = fmaxf(mx, v[c]);
        }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!PROBS) v[c] = __expf(v[c] - mx);
            sum += v[c];
        }
        const float inv = PROBS ? 1.0f : 1.0f / sum;
        const int lab = (int)labels[i];
        float G[NC], dot = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            v[c] *= inv;
            G[c] = (c == lab ? a[c] : 0.f) + bb[c] * v[c];
            dot += v[c] * G[c];
        }
        float* dp = dlogits + b * NC * HW + p;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float oh = (c == lab) ? 1.f : 0.f;
            if (PROBS) dp[c * HW] = g * dice_scale * G[c];              // d/dp directly: no softmax Jacobian (CE is not defined on probabilities here)
            else dp[c * HW] = g * (ce_scale * (v[c] - oh) + dice_scale * v[c] * (G[c] - dot));
        }
    }
}

int loss_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

#define NC_SWITCH(NCV, CALL)                                                                     \
    switch (NCV) {                                                                               \
        case 2: { constexpr int NC = 2; CALL; } break;                                           \
        case 3: { constexpr int NC = 3; CALL; } break;                                           \
        case 4: { constexpr int NC = 4; CALL; } break;                                           \
        case 5: { constexpr int NC = 5; CALL; } break;                                           \
        case 6: { constexpr int NC = 6; CALL; } break;                                           \
        case 7: { constexpr int NC = 7; CALL; } break;                                           \
        case 8: { constexpr int NC = 8; CALL; } break;                                           \
        case 9: { constexpr int NC = 9; CALL; } break;                                           \
        case 14: { constexpr int NC = 14; CALL; } break;                                         \
        case 16: { constexpr int NC = 16; CALL; } break;                                         \
        default: cswin_set_error("loss: num_classes=%d unsupported", NCV); return CSWIN_ERR_UNSUPPORTED; \
    }

}  // namespace

extern "C" {

size_t cswin_loss_workspace(int B, int ncls, long HW) { return (size_t)loss_blocks((long)B * HW) * (1 + 3 * ncls) * sizeof(float); }

// logits (B, ncls, HW) fp32, labels (B, HW) int64 -> sums[1 + 3*ncls] (local batch)
int cswin_loss_sums(const float* logits, const long long* labels, float* sums, void* workspace, size_t ws_bytes, int B,
                    int ncls, long HW, int inputs_are_probs, void* stream) {
    CSWIN_REQUIRE(logits && labels && sums && B > 0 && HW > 0, CSWIN_ERR_SHAPE, "loss_sums: bad arguments");
    CSWIN_REQUIRE(workspace && ws_bytes >= cswin_loss_workspace(B, ncls, HW), CSWIN_ERR_WORKSPACE, "loss_sums: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = loss_blocks((long)B * HW);
    if (inputs_are_probs) {
        NC_SWITCH(ncls, hipLaunchKernelGGL((loss_sums_kernel<NC, true>), dim3(nblk), dim3(256), 0, st, logits, labels, (float*)workspace, B, HW));
    } else {
        NC_SWITCH(ncls, hipLaunchKernelGGL((loss_sums_kernel<NC, false>), dim3(nblk), dim3(256), 0, st, logits, labels, (float*)workspace, B, HW));
    }
    CSWIN_LAUNCH_CHECK();
    launch_rows_sum((const float*)workspace, sums, nullptr, 0, 1 + 3 * ncls, nblk, 1 + 3 * ncls, st);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// sums (possibly all-reduced) -> out[3] = {loss, ce, dice}, coef[2*ncls]; n_pixels = pixel count the sums cover
int cswin_loss_finalize(const float* sums, float* out, float* coef, double n_pixels, int ncls, float w_ce, float w_dice,
                        const float* class_weight, void* stream) {
    CSWIN_REQUIRE(sums && out && coef && n_pixels > 0 && ncls > 0, CSWIN_ERR_SHAPE, "loss_finalize: bad arguments");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, out, coef, (float)n_pixels, ncls, w_ce, w_dice, class_weight);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// ce_scale = w_ce / n_pixels ; dice_scale = w_dice / ncls (times world size under gradient averaging)
int cswin_loss_bwd(const float* logits, const long long* labels, const float* coef, const float* grad_out, float* dlogits,
                   float ce_scale, float dice_scale, int B, int ncls, long HW, int inputs_are_probs, void* stream) {
    CSWIN_REQUIRE(logits && labels && coef && dlogits && B > 0 && HW > 0, CSWIN_ERR_SHAPE, "loss_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = loss_blocks((long)B * HW) * 4;
    if (inputs_are_probs) {
        NC_SWITCH(ncls, hipLaunchKernelGGL((loss_bwd_kernel<NC, true>), dim3(nblk), dim3(256), 0, st, logits, labels, coef, grad_out, dlogits, ce_scale, dice_scale, B, HW));
    } else {
        NC_SWITCH(ncls, hipLaunchKernelGGL((loss_bwd_kernel<NC, false>), dim3(nblk), dim3(256), 0, st, logits, labels, coef, grad_out, dlogits, ce_scale, dice_scale, B, HW));
    }
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

}  // extern "C"
