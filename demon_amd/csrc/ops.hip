// ops.hip -- the lmbspecialops-level kernels of the DeMoN hot path for gfx950.
//
// Every kernel maps one 64-lane wavefront onto 64 consecutive pixels of an image row segment so that
// all streaming loads/stores are 256-byte coalesced; per-sample camera parameters are wave-uniform.
// Arithmetic is written in the same operation order as the CPU oracle so the two agree to rounding; the per-pixel geometry
// functions are compiled without FMA contraction, so that the fused assembly kernels (which inline several of them) give the
// same bits as the chains of stand-alone op launches they replace.
//
//   depth_to_flow (+ |flow|<1 gate)  blocks_original.py:155-168
//   flow_to_depth / flow_to_depth2   blocks_original.py:344-360, v2/blocks.py:362-378
//   warp2d                           blocks_original.py:171-176, :336-339
//   leaky_relu                       helpers.py:60-63
//   replace_nonfinite                v2/losses.py:49
//   scale_invariant_gradient         v2/losses.py:76-79
//   median3x3_downsample             examples/evaluation.py:173, v2/helpers.py:102
//   depth_to_normals                 v2/losses.py:336-337
#include <stdlib.h>

#include "internal.h"

namespace demon {

// angle-axis -> rotation matrix, helpers.py:37-57 (identity when angle <= 1e-6)
__device__ __forceinline__ void angleaxis_to_rotation(const float *__restrict__ aa, float R[9])
{
#pragma clang fp contract(off)  // same roundings wherever this is inlined (and as the oracle's plain C): see the header comment
    const float ax = aa[0], ay = aa[1], az = aa[2];
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    if (angle > 1e-6f) {
        const float c = cosf(angle), s = sinf(angle);
        const float ux = ax / angle, uy = ay / angle, uz = az / angle;
        const float omc = 1.0f - c;
        R[0] = c + ux * ux * omc;      R[1] = ux * uy * omc - uz * s; R[2] = ux * uz * omc + uy * s;
        R[3] = uy * ux * omc + uz * s; R[4] = c + uy * uy * omc;      R[5] = uy * uz * omc - ux * s;
        R[6] = uz * ux * omc - uy * s; R[7] = uz * uy * omc + ux * s; R[8] = c + uz * uz * omc;
    } else {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    }
}

// per-sample camera of the geometry ops: intrinsics in pixels, rotation matrix, translation
struct Camera {
    float fx, fy, cx, cy, ifx, ify;
    float R[9];
    float t[3];
};
__device__ __forceinline__ Camera load_camera(const float *__restrict__ intrinsics, const float *__restrict__ rotation,
                                              const float *__restrict__ translation, int n, int H, int W)
{
#pragma clang fp contract(off)  // same roundings wherever this is inlined (and as the oracle's plain C): see the header comment
    Camera c;
    const float *K = intrinsics + 4 * n;
    c.fx = K[0] * W; c.fy = K[1] * H; c.cx = K[2] * W; c.cy = K[3] * H;
    c.ifx = 1.0f / c.fx; c.ify = 1.0f / c.fy;
    angleaxis_to_rotation(rotation + 3 * n, c.R);
    c.t[0] = translation[3 * n]; c.t[1] = translation[3 * n + 1]; c.t[2] = translation[3 * n + 2];
    return c;
}

// sops.depth_to_flow for one pixel (+ the |flow| < 1 / NaN gate of blocks_original.py:163-168 when gate != 0)
__device__ __forceinline__ void depth_to_flow_pixel(const Camera &c, float d, int x, int y, int H, int W, int inverse_depth,
                                                    int normalize_flow, int gate, float &fxo, float &fyo)
{
#pragma clang fp contract(off)  // same roundings wherever this is inlined (and as the oracle's plain C): see the header comment
    fxo = __builtin_nanf("");
    fyo = __builtin_nanf("");
    if (inverse_depth) d = 1.0f / d;
    if (d > 0.0f && isfinite(d)) {
        const float px = x + 0.5f, py = y + 0.5f;
        const float X = d * ((px - c.cx) * c.ifx), Y = d * ((py - c.cy) * c.ify), Z = d;
        const float X2 = c.R[0] * X + c.R[1] * Y + c.R[2] * Z + c.t[0];
        const float Y2 = c.R[3] * X + c.R[4] * Y + c.R[5] * Z + c.t[1];
        const float Z2 = c.R[6] * X + c.R[7] * Y + c.R[8] * Z + c.t[2];
        const float p2x = c.fx * X2 / Z2 + c.cx, p2y = c.fy * Y2 / Z2 + c.cy;
        fxo = p2x - px;
        fyo = p2y - py;
        if (normalize_flow) { fxo /= W; fyo /= H; }
    }
    if (gate) {
        const float nrm = sqrtf(fxo * fxo + fyo * fyo);
        if (!(nrm < 1.0f)) { fxo = 0.0f; fyo = 0.0f; }
    }
}

// grid: (ceil(H*W/256), N)
__global__ __launch_bounds__(256) void depth_to_flow_kernel(float *__restrict__ out, const float *__restrict__ depth,
                                                            long depth_n_stride, const float *__restrict__ intrinsics,
                                                            const float *__restrict__ rotation,
                                                            const float *__restrict__ translation, int H, int W,
                                                            long out_n_stride, int inverse_depth, int normalize_flow,
                                                            int gate)
{
    const int n = blockIdx.y;
    const int hw = H * W;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int y = idx / W, x = idx - y * W;
    const Camera c = load_camera(intrinsics, rotation, translation, n, H, W);
    float fxo, fyo;
    depth_to_flow_pixel(c, depth[(long)n * depth_n_stride + idx], x, y, H, W, inverse_depth, normalize_flow, gate, fxo, fyo);
    float *o = out + (long)n * out_n_stride + idx;
    o[0] = fxo;
    o[hw] = fyo;
}

// One-sided Jacobi SVD of a 4x4 matrix, 8 fixed sweeps; returns the right singular vector of the
// smallest singular value.  Same operation sequence as jacobi_null4 in oracle/demon_oracle.c.
__device__ __forceinline__ void jacobi_null4(float A[4][4], float X[4])
{
#pragma clang fp contract(off)  // same roundings wherever this is inlined (and as the oracle's plain C): see the header comment
    float V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                float alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    alpha += A[i][p] * A[i][p];
                    beta += A[i][q] * A[i][q];
                    gamma += A[i][p] * A[i][q];
                }
                const bool skip = fabsf(gamma) <= 1e-30f || !(fabsf(gamma) > 1e-12f * sqrtf(alpha * beta));
                const float zeta = (beta - alpha) / (2.0f * gamma);
                const float tt = (zeta >= 0 ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                float c = 1.0f / sqrtf(1.0f + tt * tt), s = c * tt;
                if (skip) { c = 1.0f; s = 0.0f; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ap = A[i][p], aq = A[i][q];
                    A[i][p] = skip ? ap : c * ap - s * aq;
                    A[i][q] = skip ? aq : s * ap + c * aq;
                    const float vp = V[i][p], vq = V[i][q];
                    V[i][p] = skip ? vp : c * vp - s * vq;
                    V[i][q] = skip ? vq : s * vp + c * vq;
                }
            }
    }
    float bestn = __builtin_inff();
    X[0] = X[1] = X[2] = X[3] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float nn = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) nn += A[i][j] * A[i][j];
        if (nn < bestn) {
            bestn = nn;
#pragma unroll
            for (int i = 0; i < 4; ++i) X[i] = V[i][j];
        }
    }
}

// sops.flow_to_depth (method 0: DLT / SVD) / flow_to_depth2 (method 1: closed form) for one pixel
__device__ __forceinline__ float flow_to_depth_pixel(const Camera &c, float u, float v, int x, int y, int H, int W, int inverse_depth,
                                                     int normalized_flow, int method, float clip_hi)
{
#pragma clang fp contract(off)  // same roundings wherever this is inlined (and as the oracle's plain C): see the header comment
    const float fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    const float *R = c.R, *t = c.t;
    if (normalized_flow) { u *= W; v *= H; }
    const float p1x = x + 0.5f, p1y = y + 0.5f;
    const float p2x = p1x + u, p2y = p1y + v;
    float z;
    if (method == 0) {
        float P2[3][4];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            P2[0][j] = fx * R[0 + j] + cx * R[6 + j];
            P2[1][j] = fy * R[3 + j] + cy * R[6 + j];
            P2[2][j] = R[6 + j];
        }
        P2[0][3] = fx * t[0] + cx * t[2];
        P2[1][3] = fy * t[1] + cy * t[2];
        P2[2][3] = t[2];
        float A[4][4];
        A[0][0] = -fx; A[0][1] = 0;   A[0][2] = p1x - cx; A[0][3] = 0;
        A[1][0] = 0;   A[1][1] = -fy; A[1][2] = p1y - cy; A[1][3] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            A[2][j] = p2x * P2[2][j] - P2[0][j];
            A[3][j] = p2y * P2[2][j] - P2[1][j];
        }
        float X[4];
        jacobi_null4(A, X);
        z = X[2] / X[3];
    } else {
        const float rx = (p1x - cx) / fx, ry = (p1y - cy) / fy;
        const float qx = R[0] * rx + R[1] * ry + R[2];
        const float qy = R[3] * rx + R[4] * ry + R[5];
        const float qz = R[6] * rx + R[7] * ry + R[8];
        const float ax = fx * qx - (p2x - cx) * qz, bx = (p2x - cx) * t[2] - fx * t[0];
        const float ay = fy * qy - (p2y - cy) * qz, by = (p2y - cy) * t[2] - fy * t[1];
        z = (ax * bx + ay * by) / (ax * ax + ay * ay);
    }
    float r = inverse_depth ? 1.0f / z : z;
    // v2/blocks.py:379 tf.clip_by_value(., 0, 50) fused here (clip_hi > 0): max(min(r, hi), 0) with fminf / fmaxf, i.e. the
    // semantics of TF's GPU kernels, the only backend the v2 driver accepts (example_v2.py:50-54): NaN -> hi
    if (clip_hi > 0.0f) r = fmaxf(fminf(r, clip_hi), 0.0f);
    return r;
}

// grid: (ceil(H*W/256), N)
__global__ __launch_bounds__(256) void flow_to_depth_kernel(float *__restrict__ out, long out_n_stride,
                                                            const float *__restrict__ flow, long flow_n_stride,
                                                            const float *__restrict__ intrinsics,
                                                            const float *__restrict__ rotation,
                                                            const float *__restrict__ translation, int H, int W,
                                                            int inverse_depth, int normalized_flow, int method,
                                                            float clip_hi)
{
    const int n = blockIdx.y;
    const int hw = H * W;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int y = idx / W, x = idx - y * W;
    const Camera c = load_camera(intrinsics, rotation, translation, n, H, W);
    const float u = flow[(long)n * flow_n_stride + idx];
    const float v = flow[(long)n * flow_n_stride + hw + idx];
    out[(long)n * out_n_stride + idx] = flow_to_depth_pixel(c, u, v, x, y, H, W, inverse_depth, normalized_flow, method, clip_hi);
}

// sops.warp2d for one pixel: the four bilinear taps (clamped indices, inside flags, weights) decided once per pixel ...
struct WarpTaps {
    int xi[2], yi[2];
    bool okx[2], oky[2], finite;
    float w00, w01, w10, w11;
};
__device__ __forceinline__ WarpTaps warp_taps(float dx, float dy, int x, int y, int H, int W, int normalized)
{
#pragma clang fp contract(off)  // same roundings wherever this is inlined (and as the oracle's plain C): see the header comment
    WarpTaps t;
    if (normalized) { dx *= W; dy *= H; }
    const float sx = x + dx, sy = y + dy;
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const float a = sx - fx0, b = sy - fy0;
    t.finite = isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
    const int x0 = t.finite ? (int)fx0 : -2, y0 = t.finite ? (int)fy0 : -2;
    t.w00 = (1.0f - a) * (1.0f - b); t.w01 = a * (1.0f - b); t.w10 = (1.0f - a) * b; t.w11 = a * b;
    t.xi[0] = x0; t.xi[1] = x0 + 1; t.yi[0] = y0; t.yi[1] = y0 + 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        t.okx[k] = t.xi[k] >= 0 && t.xi[k] < W;
        t.oky[k] = t.yi[k] >= 0 && t.yi[k] < H;
        t.xi[k] = min(max(t.xi[k], 0), W - 1);
        t.yi[k] = min(max(t.yi[k], 0), H - 1);
    }
    return t;
}
// ... and applied to every channel plane p
__device__ __forceinline__ float warp_plane(const WarpTaps &t, const float *__restrict__ p, int W, bool value_mode, float border_value)
{
#pragma clang fp contract(off)  // same roundings wherever this is inlined (and as the oracle's plain C): see the header comment
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int kx = k & 1, ky = k >> 1;
        const float g = p[t.yi[ky] * W + t.xi[kx]];
        v[k] = (value_mode && !(t.finite && t.okx[kx] && t.oky[ky])) ? border_value : g;
    }
    if (t.finite) return t.w00 * v[0] + t.w01 * v[1] + t.w10 * v[2] + t.w11 * v[3];
    return value_mode ? border_value : __builtin_nanf("");
}

// Backward bilinear warp by direct gather.  grid: (ceil(H*W/256), N); each thread handles one pixel for all C channels; the four
// taps of neighbouring lanes fall into the same or adjacent 128-byte lines.  The source planes (3 x 48 x 64 floats = 36 KB per
// pair at 256x192, 230 KB at 640x480) sit in L1 / L2, which is why staging them through LDS does not pay: a staged variant
// (4 x 64 output tile, block-reduced bounding box of the taps copied to LDS, direct-gather fallback for large or incoherent
// displacements) measured 82-100 us against 18-25 us for this kernel on the 120 x 160 level-2 maps of the 640x480 workload
// (batch 64, round 1) and was removed in round 2; at 48 x 64 either is bound by the launch.
__global__ __launch_bounds__(256) void warp2d_kernel(float *__restrict__ out, long out_n_stride,
                                                     const float *__restrict__ in, long in_n_stride,
                                                     const float *__restrict__ disp, long disp_n_stride, int C,
                                                     int H, int W, int normalized, int border_mode,
                                                     float border_value)
{
    const int n = blockIdx.y;
    const int hw = H * W;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int y = idx / W, x = idx - y * W;
    const WarpTaps t = warp_taps(disp[(long)n * disp_n_stride + idx], disp[(long)n * disp_n_stride + hw + idx], x, y, H, W, normalized);
    for (int c = 0; c < C; ++c)
        out[(long)n * out_n_stride + (long)c * hw + idx] = warp_plane(t, in + (long)n * in_n_stride + (long)c * hw, W, border_mode == 1, border_value);
}

// ---- fused assembly of the extra inputs of the iterative blocks (one launch instead of three or four) ----------------------
// Flow block (blocks_original.py:155-183; v2/blocks.py:154-183): extra = [image2_2 warped by the flow that the previous depth
// and motion imply (3), that flow, gated (2), previous depth (1), previous normals (3)].  `dn` is the 4-channel depth+normal
// buffer of the previous stage.  Same per-pixel functions as the stand-alone kernels, so the result is bit-identical.
// grid: (ceil(H*W/256), N)
__global__ __launch_bounds__(256) void assemble_flow_inputs_kernel(float *__restrict__ extra, long extra_n_stride,
                                                                   const float *__restrict__ img2, long img2_n_stride,
                                                                   const float *__restrict__ dn, long dn_n_stride,
                                                                   const float *__restrict__ intrinsics, const float *__restrict__ rotation,
                                                                   const float *__restrict__ translation, int H, int W)
{
    const int n = blockIdx.y;
    const int hw = H * W;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int y = idx / W, x = idx - y * W;
    const Camera c = load_camera(intrinsics, rotation, translation, n, H, W);
    const float *__restrict__ d4 = dn + (long)n * dn_n_stride + idx;
    const float d = d4[0];
    float fx, fy;
    depth_to_flow_pixel(c, d, x, y, H, W, 1, 1, 1, fx, fy);
    float *__restrict__ o = extra + (long)n * extra_n_stride + idx;
    const WarpTaps t = warp_taps(fx, fy, x, y, H, W, 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[(long)ch * hw] = warp_plane(t, img2 + (long)n * img2_n_stride + (long)ch * hw, W, true, 0.0f);
    o[3l * hw] = fx;
    o[4l * hw] = fy;
    o[5l * hw] = d;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[(long)(6 + ch) * hw] = d4[(long)(1 + ch) * hw];
}

// Depth+motion block (blocks_original.py:335-362; v2/blocks.py:353-381): extra = [image2_2 warped by the predicted flow (3),
// flow + confidence (4), and in the iterative net the depth triangulated from that flow and the previous motion (1)].
__global__ __launch_bounds__(256) void assemble_dm_inputs_kernel(float *__restrict__ extra, long extra_n_stride,
                                                                 const float *__restrict__ img2, long img2_n_stride,
                                                                 const float *__restrict__ flowconf, long fc_n_stride,
                                                                 const float *__restrict__ intrinsics, const float *__restrict__ rotation,
                                                                 const float *__restrict__ translation, int H, int W, int with_depth,
                                                                 int method, float clip_hi)
{
    const int n = blockIdx.y;
    const int hw = H * W;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int y = idx / W, x = idx - y * W;
    const float *__restrict__ fc = flowconf + (long)n * fc_n_stride + idx;
    const float u = fc[0], v = fc[hw];
    float *__restrict__ o = extra + (long)n * extra_n_stride + idx;
    const WarpTaps t = warp_taps(u, v, x, y, H, W, 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[(long)ch * hw] = warp_plane(t, img2 + (long)n * img2_n_stride + (long)ch * hw, W, true, 0.0f);
    o[3l * hw] = u;
    o[4l * hw] = v;
    o[5l * hw] = fc[2l * hw];
    o[6l * hw] = fc[3l * hw];
    if (with_depth) {
        const Camera c = load_camera(intrinsics, rotation, translation, n, H, W);
        o[7l * hw] = flow_to_depth_pixel(c, u, v, x, y, H, W, 1, 1, method, clip_hi);
    }
}

// refinement net input (blocks_original.py:475-482): [image1 (3), depth2 upsampled x4 by nearest neighbour (1)] in one launch
// grid: (ceil(H*W/256), N); H, W = full resolution
__global__ __launch_bounds__(256) void assemble_refine_input_kernel(float *__restrict__ out, long out_n_stride,
                                                                    const float *__restrict__ image, long image_n_stride,
                                                                    const float *__restrict__ depth2, long depth_n_stride, int H, int W, int factor)
{
    const int n = blockIdx.y;
    const int hw = H * W;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int y = idx / W, x = idx - y * W;
    float *__restrict__ o = out + (long)n * out_n_stride + idx;
    const float *__restrict__ im = image + (long)n * image_n_stride + idx;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[(long)ch * hw] = im[(long)ch * hw];
    o[3l * hw] = depth2[(long)n * depth_n_stride + (y / factor) * (W / factor) + (x / factor)];
}

__global__ __launch_bounds__(256) void leaky_relu_kernel(float *__restrict__ out, const float *__restrict__ in,
                                                         long count, float leak)
{
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < count) {
        float4 v = *reinterpret_cast<const float4 *>(in + i4);
        v.x = v.x >= 0.0f ? v.x : leak * v.x;
        v.y = v.y >= 0.0f ? v.y : leak * v.y;
        v.z = v.z >= 0.0f ? v.z : leak * v.z;
        v.w = v.w >= 0.0f ? v.w : leak * v.w;
        *reinterpret_cast<float4 *>(out + i4) = v;
    } else {
        for (long i = i4; i < count; ++i) out[i] = in[i] >= 0.0f ? in[i] : leak * in[i];
    }
}

__global__ __launch_bounds__(256) void replace_nonfinite_kernel(float *__restrict__ out, const float *__restrict__ in,
                                                                long count, float value)
{
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < count) {
        float4 v = *reinterpret_cast<const float4 *>(in + i4);
        v.x = isfinite(v.x) ? v.x : value;
        v.y = isfinite(v.y) ? v.y : value;
        v.z = isfinite(v.z) ? v.z : value;
        v.w = isfinite(v.w) ? v.w : value;
        *reinterpret_cast<float4 *>(out + i4) = v;
    } else {
        for (long i = i4; i < count; ++i) out[i] = isfinite(in[i]) ? in[i] : value;
    }
}

// scale invariant gradient; block = 64 x 4 pixel tile, LDS tile with a halo of max|delta| columns/rows.
// grid: (ceil(W/64), ceil(H/4), NC)
#define SIG_MAX_DELTAS 8
struct SigParams { int deltas[SIG_MAX_DELTAS]; float weights[SIG_MAX_DELTAS]; int n; };
#define SIG_MAX_HALO 16
__global__ __launch_bounds__(256) void sig_kernel(float *__restrict__ out, const float *__restrict__ in, int H, int W,
                                                  SigParams sp, float eps, int halo)
{
    // 4 x 64 tile plus `halo` = max |delta| pixels on every side, staged once; all neighbours come from LDS
    __shared__ float tile[(4 + 2 * SIG_MAX_HALO) * (64 + 2 * SIG_MAX_HALO)];
    const int z = blockIdx.z;
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * 4;
    const float *p = in + (long)z * H * W;
    const bool staged = halo <= SIG_MAX_HALO;
    const int tw = 64 + 2 * halo, th = 4 + 2 * halo;
    if (staged) {
        for (int e = threadIdx.x; e < tw * th; e += 256) {
            const int ry = e / tw, rx = e - ry * tw;
            const int gy = ty0 - halo + ry, gx = tx0 - halo + rx;
            tile[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? p[gy * W + gx] : 0.0f;
        }
        __syncthreads();
    }
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) return;
    auto at = [&](int yy, int xx) { return staged ? tile[(yy - ty0 + halo) * tw + (xx - tx0 + halo)] : p[yy * W + xx]; };
    const float u = at(y, x);
    // lmbspecialops contract: channels fold into the batch ([N,C,H,W] -> [N*C,2,H,W]) and the deltas are SUMMED with their weights;
    // that is why v2/losses.py:76-79 calls the op once per delta and concatenates the results itself
    float gx = 0.0f, gy = 0.0f;
    for (int k = 0; k < sp.n; ++k) {
        const int d = sp.deltas[k];
        if (x + d >= 0 && x + d < W) {
            const float un = at(y, x + d);
            gx += sp.weights[k] * (un - u) / (fabsf(un) + fabsf(u) + eps);
        }
        if (y + d >= 0 && y + d < H) {
            const float un = at(y + d, x);
            gy += sp.weights[k] * (un - u) / (fabsf(un) + fabsf(u) + eps);
        }
    }
    out[((long)z * 2 + 0) * H * W + y * W + x] = gx;
    out[((long)z * 2 + 1) * H * W + y * W + x] = gy;
}

// Median keys: a monotone map of float bits to unsigned (negative values reversed, sign bit flipped) with every NaN mapped
// behind +inf, so that integer min / max sort in the oracle's total order ("NaN sorts last", oracle/demon_oracle.c cmp_float).
__device__ __forceinline__ unsigned median_key(float v)
{
    const unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float median_unkey(unsigned k)
{
    if (k == 0xffffffffu) return __builtin_nanf("");
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ void cswap(unsigned &a, unsigned &b)
{
    const unsigned lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}
// grid: (ceil(Wo/64), ceil(Ho/4), NC)
__global__ __launch_bounds__(256) void median3x3_downsample_kernel(float *__restrict__ out, const float *__restrict__ in,
                                                                   int H, int W, int Ho, int Wo)
{
    const int z = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wo || y >= Ho) return;
    const float *p = in + (long)z * H * W;
    unsigned v[9];
    int k = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = min(max(2 * y + dy, 0), H - 1), xx = min(max(2 * x + dx, 0), W - 1);
            v[k++] = median_key(p[yy * W + xx]);
        }
    // median-of-9 network (Paeth)
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
    cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
    cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
    cswap(v[4], v[2]);
    out[(long)z * Ho * Wo + y * Wo + x] = median_unkey(v[4]);
}

// sops.depth_to_normals (v2/losses.py:336-337): camera-frame normals from a depth map; same operation order as
// ref_depth_to_normals in oracle/demon_oracle.c (semantics documented there; unpinned).
// grid: (ceil(W/64), ceil(H/4), N)
__global__ __launch_bounds__(256) void depth_to_normals_kernel(float *__restrict__ out, const float *__restrict__ depth,
                                                               const float *__restrict__ intrinsics, int H, int W,
                                                               int inverse_depth)
{
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int hw = H * W;
    const float *K = intrinsics + 4 * n;
    const float fx = K[0] * W, fy = K[1] * H, cx = K[2] * W, cy = K[3] * H;
    const float ifx = 1.0f / fx, ify = 1.0f / fy;
    const float *D = depth + (long)n * hw;
    float nx = __builtin_nanf(""), ny = nx, nz = nx;
    if (x > 0 && y > 0 && x < W - 1 && y < H - 1) {
        const int xs[5] = {x, x - 1, x + 1, x, x}, ys[5] = {y, y, y, y - 1, y + 1};
        float P[5][3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float d = D[ys[k] * W + xs[k]];
            if (inverse_depth) d = 1.0f / d;
            if (!(d > 0.0f) || !isfinite(d)) ok = false;
            P[k][0] = d * ((xs[k] + 0.5f - cx) * ifx);
            P[k][1] = d * ((ys[k] + 0.5f - cy) * ify);
            P[k][2] = d;
        }
        if (ok) {
            float dx[3], dy[3];
            const bool bx = fabsf(P[0][2] - P[1][2]) < fabsf(P[2][2] - P[0][2]);
            const bool by = fabsf(P[0][2] - P[3][2]) < fabsf(P[4][2] - P[0][2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dx[c] = bx ? P[0][c] - P[1][c] : P[2][c] - P[0][c];
                dy[c] = by ? P[0][c] - P[3][c] : P[4][c] - P[0][c];
            }
            const float c0 = dy[1] * dx[2] - dy[2] * dx[1];
            const float c1 = dy[2] * dx[0] - dy[0] * dx[2];
            const float c2 = dy[0] * dx[1] - dy[1] * dx[0];
            const float inv = 1.0f / sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
            nx = c0 * inv; ny = c1 * inv; nz = c2 * inv;
        }
    }
    float *o = out + (long)n * 3 * hw + y * W + x;
    o[0] = nx;
    o[hw] = ny;
    o[2 * hw] = nz;
}

// grid: (ceil(C*HW/1024), N)
__global__ __launch_bounds__(256) void copy_channels_kernel(float *__restrict__ dst, long dst_n_stride,
                                                            const float *__restrict__ src, long src_n_stride,
                                                            long chw)
{
    const int n = blockIdx.y;
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const float *s = src + (long)n * src_n_stride;
    float *d = dst + (long)n * dst_n_stride;
    if (i4 + 3 < chw) {
        *reinterpret_cast<float4 *>(d + i4) = *reinterpret_cast<const float4 *>(s + i4);
    } else {
        for (long i = i4; i < chw; ++i) d[i] = s[i];
    }
}

// tf.image.resize_nearest_neighbor for integer factors (blocks_original.py:475): src = dst / factor
// grid: (ceil(Ho*Wo/256), C, N)
__global__ __launch_bounds__(256) void upsample_nearest_kernel(float *__restrict__ dst, long dst_n_stride,
                                                               const float *__restrict__ src, long src_n_stride,
                                                               int H, int W, int factor)
{
    const int n = blockIdx.z, c = blockIdx.y;
    const int Ho = H * factor, Wo = W * factor;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Ho * Wo) return;
    const int y = idx / Wo, x = idx - y * Wo;
    dst[(long)n * dst_n_stride + (long)c * Ho * Wo + idx] =
        src[(long)n * src_n_stride + (long)c * H * W + (y / factor) * W + (x / factor)];
}

__global__ void split_motion_kernel(const float *__restrict__ motion, float *__restrict__ rot,
                                    float *__restrict__ trans, float *__restrict__ scale, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 7) return;
    const int n = i / 7, k = i - n * 7;
    const float v = motion[i];
    if (k < 3) rot[n * 3 + k] = v;
    else if (k < 6) trans[n * 3 + k - 3] = v;
    else scale[n] = v;
}

// ---- launchers ----------------------------------------------------------------------------------
void launch_depth_to_flow(float *out, const float *depth, long depth_n_stride, const float *intrinsics,
                          const float *rotation, const float *translation, int N, int H, int W,
                          long out_n_stride, int inverse_depth, int normalize_flow, int gate, hipStream_t s)
{
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(depth_to_flow_kernel, grid, dim3(256), 0, s, out, depth, depth_n_stride, intrinsics, rotation,
                       translation, H, W, out_n_stride, inverse_depth, normalize_flow, gate);
}

void launch_flow_to_depth(float *out, long out_n_stride, const float *flow, long flow_n_stride,
                          const float *intrinsics, const float *rotation, const float *translation, int N, int H,
                          int W, int inverse_depth, int normalized_flow, int method, float clip_hi, hipStream_t s)
{
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(flow_to_depth_kernel, grid, dim3(256), 0, s, out, out_n_stride, flow, flow_n_stride, intrinsics,
                       rotation, translation, H, W, inverse_depth, normalized_flow, method, clip_hi);
}

void launch_warp2d(float *out, long out_n_stride, const float *in, long in_n_stride, const float *disp,
                   long disp_n_stride, int N, int C, int H, int W, int normalized, int border_mode,
                   float border_value, hipStream_t s)
{
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(warp2d_kernel, grid, dim3(256), 0, s, out, out_n_stride, in, in_n_stride, disp, disp_n_stride, C,
                       H, W, normalized, border_mode, border_value);
}

void launch_assemble_flow_inputs(float *extra, long extra_n_stride, const float *img2, long img2_n_stride, const float *dn,
                                 long dn_n_stride, const float *intrinsics, const float *rotation, const float *translation, int N, int H,
                                 int W, hipStream_t s)
{
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(assemble_flow_inputs_kernel, grid, dim3(256), 0, s, extra, extra_n_stride, img2, img2_n_stride, dn, dn_n_stride,
                       intrinsics, rotation, translation, H, W);
}

void launch_assemble_dm_inputs(float *extra, long extra_n_stride, const float *img2, long img2_n_stride, const float *flowconf,
                               long fc_n_stride, const float *intrinsics, const float *rotation, const float *translation, int N, int H,
                               int W, int with_depth, int method, float clip_hi, hipStream_t s)
{
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(assemble_dm_inputs_kernel, grid, dim3(256), 0, s, extra, extra_n_stride, img2, img2_n_stride, flowconf, fc_n_stride,
                       intrinsics, rotation, translation, H, W, with_depth, method, clip_hi);
}

void launch_assemble_refine_input(float *out, long out_n_stride, const float *image, long image_n_stride, const float *depth2,
                                  long depth_n_stride, int N, int H, int W, int factor, hipStream_t s)
{
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(assemble_refine_input_kernel, grid, dim3(256), 0, s, out, out_n_stride, image, image_n_stride, depth2, depth_n_stride, H, W,
                       factor);
}

void launch_leaky_relu(float *out, const float *in, long count, float leak, hipStream_t s)
{
    const long blocks = (count + 1023) / 1024;
    hipLaunchKernelGGL(leaky_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, in, count, leak);
}

void launch_replace_nonfinite(float *out, const float *in, long count, float value, hipStream_t s)
{
    const long blocks = (count + 1023) / 1024;
    hipLaunchKernelGGL(replace_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, in, count, value);
}

void launch_sig(float *out, const float *in, int NC, int H, int W, const int *deltas, const float *weights,
                int ndeltas, float eps, hipStream_t s)
{
    SigParams sp;
    sp.n = ndeltas;
    for (int i = 0; i < ndeltas; ++i) { sp.deltas[i] = deltas[i]; sp.weights[i] = weights[i]; }
    dim3 grid((W + 63) / 64, (H + 3) / 4, NC);
    int halo = 0;
    for (int i = 0; i < ndeltas; ++i) halo = deltas[i] < 0 ? (halo > -deltas[i] ? halo : -deltas[i]) : (halo > deltas[i] ? halo : deltas[i]);
    hipLaunchKernelGGL(sig_kernel, grid, dim3(256), 0, s, out, in, H, W, sp, eps, halo);
}

// pointwise_l2_loss (v2/losses.py:33-54): mean over (n, y, x) of sqrt(sum_c replace_nonfinite(inp - gt)^2 + epsilon).
// grid: ceil(N*H*W / 256) blocks; every block writes the sum of its 256 pixels to partial[blockIdx.x]
// (wave shuffle + LDS reduction); the host adds the few hundred partials in double.
__global__ __launch_bounds__(256) void pointwise_l2_partial_kernel(float *__restrict__ partial, const float *__restrict__ inp,
                                                                   const float *__restrict__ gt, int N, int C, int HW, float epsilon)
{
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    float v = 0.0f;
    if (p < (long)N * HW) {
        const int n = (int)(p / HW), idx = (int)(p - (long)n * HW);
        const long base = (long)n * C * HW + idx;
        float s = 0.0f;
        for (int c = 0; c < C; ++c) {
            float d = inp[base + (long)c * HW] - gt[base + (long)c * HW];
            if (!isfinite(d)) d = 0.0f;  // sops.replace_nonfinite(inp - gt), value 0
            s = fmaf(d, d, s);
        }
        v = sqrtf(s + epsilon);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

void launch_pointwise_l2_partial(float *partial, const float *inp, const float *gt, int N, int C, int HW, float epsilon, hipStream_t s)
{
    const long P = (long)N * HW;
    hipLaunchKernelGGL(pointwise_l2_partial_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, partial, inp, gt, N, C, HW, epsilon);
}

void launch_median3x3_downsample(float *out, const float *in, int NC, int H, int W, hipStream_t s)
{
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    dim3 grid((Wo + 63) / 64, (Ho + 3) / 4, NC);
    hipLaunchKernelGGL(median3x3_downsample_kernel, grid, dim3(256), 0, s, out, in, H, W, Ho, Wo);
}

void launch_depth_to_normals(float *out, const float *depth, const float *intrinsics, int N, int H, int W, int inverse_depth,
                             hipStream_t s)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4, N);
    hipLaunchKernelGGL(depth_to_normals_kernel, grid, dim3(256), 0, s, out, depth, intrinsics, H, W, inverse_depth);
}

void launch_copy_channels(float *dst, long dst_n_stride, const float *src, long src_n_stride, int N, int C, long HW,
                          hipStream_t s)
{
    const long chw = (long)C * HW;
    dim3 grid((unsigned)((chw + 1023) / 1024), N);
    hipLaunchKernelGGL(copy_channels_kernel, grid, dim3(256), 0, s, dst, dst_n_stride, src, src_n_stride, chw);
}

void launch_upsample_nearest(float *dst, long dst_n_stride, const float *src, long src_n_stride, int N, int C, int H,
                             int W, int factor, hipStream_t s)
{
    dim3 grid((H * factor * W * factor + 255) / 256, C, N);
    hipLaunchKernelGGL(upsample_nearest_kernel, grid, dim3(256), 0, s, dst, dst_n_stride, src, src_n_stride, H, W,
                       factor);
}

void launch_split_motion(const float *motion, float *rot, float *trans, float *scale, int N, hipStream_t s)
{
    hipLaunchKernelGGL(split_motion_kernel, dim3((N * 7 + 63) / 64), dim3(64), 0, s, motion, rot, trans, scale, N);
}

}  // namespace demon
