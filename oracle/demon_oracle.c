/*
 * demon_oracle.c -- CPU restatement (plain C, scalar loops) of the DeMoN inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (demon_amd/, libdemon_hip.so) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in two third-party dependencies that are
 * absent from /root/reference -- tensorflow 1.4.0 (Dockerfile:14; conv2d / conv2d_transpose /
 * dense) and lmbspecialops at an unknown commit (.gitmodules:1-3, empty submodule; depth_to_flow,
 * flow_to_depth, warp2d, leaky_relu, ...).  The reference ships no tests or golden vectors for
 * this path.  What IS pinned, against the reference's own in-tree code compiled
 * and run here by oracle/build_ref.py with the answers stored in tests/golden/sculpture_geometry.npz:
 *   - depth_to_flow: its OUTPUT (flow in pixels, NaN at invalid depth) equals the reference's
 *     flow-from-depth routine dataset_tools/view_tools_cython.pyx:196-244 on the sculpture pair, and its
 *     geometry (pixel centre x+0.5, X2 = R*X1 + t, project with K) the visibility / depth-ratio
 *     routines :9-59, :108-161;
 *   - flow_to_depth (both methods): applied to that reference flow they return the reference's depth;
 *   - the angle-axis convention of python/depthmotionnet/helpers.py:37-57.
 *   - a SECOND restatement of the depth -> flow geometry, from a different reference file written in the world frame
 *     (multivih5datareaderop/multivih5datareader.cpp:369-424 / :431-501 -> oracle/reader_ref.py), agrees with that golden flow,
 *     with the reference's visibility mask and with ref_depth_to_flow fed the relative motion (tests/test_pins.py);
 *   - warp2d's displacement SIGN, CHANNEL ORDER (0 = x) and NORMALISATION (normalized: flow / (W, H)): pulling sculpture image 2
 *     back by the reference flow reproduces image 1 on the reference's visible mask (NCC 0.64 vs 0.25 unwarped, 0.06 negated);
 *     the sub-pixel convention (sample at index x + dx) is pinned only by the zero-displacement identity -- the photo pair is
 *     too noisy to separate half-pixel shifts (tests/test_pins.py, tests/test_ops_gpu.py::test_photometric_warp_kat_hip);
 *   - prepare_input_data (examples/example.py:15-42) is pinned bit for bit to the reference function run here
 *     (tests/golden/make_golden_inputs.py, tests/test_preprocess.py) -- host code, listed for completeness.
 * Still unpinned (no in-tree material constrains them; each function states the rule it implements):
 *   warp2d's treatment of taps outside the image and of non-finite displacements, flow_to_depth on geometrically
 *   inconsistent flow (DLT vs closed form), scale_invariant_gradient (border rule), median3x3_downsample
 *   (NaN ordering: "NaN sorts last" here), depth_to_normals (difference scheme, orientation), and every TensorFlow layer
 *   (conv2d 'valid' on a padded input, conv2d_transpose, dense, 'same' padding of the v2 model; checked against PyTorch and
 *   naive loops only): they restate published semantics at the reference's call sites, which each function cites.
 *
 * All tensors are NCHW float32, contiguous.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * angle-axis -> rotation matrix.  Follows python/depthmotionnet/helpers.py:37-57
 * (angle = |aa|; identity when angle <= 1e-6; Rodrigues otherwise), in float32.
 * R is row-major 3x3.
 * ------------------------------------------------------------------------------------------- */
void ref_angleaxis_to_rotation(const float *aa, float *R)
{
    float angle = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    if (angle > 1e-6f) {
        float c = cosf(angle), s = sinf(angle);
        float ux = aa[0] / angle, uy = aa[1] / angle, uz = aa[2] / angle;
        float omc = 1.0f - c;
        R[0] = c + ux * ux * omc;      R[1] = ux * uy * omc - uz * s; R[2] = ux * uz * omc + uy * s;
        R[3] = uy * ux * omc + uz * s; R[4] = c + uy * uy * omc;      R[5] = uy * uz * omc - ux * s;
        R[6] = uz * ux * omc - uy * s; R[7] = uz * uy * omc + ux * s; R[8] = c + uz * uz * omc;
    } else {
        R[0] = 1; R[1] = 0; R[2] = 0;
        R[3] = 0; R[4] = 1; R[5] = 0;
        R[6] = 0; R[7] = 0; R[8] = 1;
    }
}

/* ---------------------------------------------------------------------------------------------
 * sops.depth_to_flow -- call site python/depthmotionnet/blocks_original.py:155-162
 * (inverse_depth=True, normalize_flow=True).  Geometry pinned by the in-tree restatements
 * dataset_tools/view_tools_cython.pyx:196-240 and multivih5datareader.cpp:369-424:
 * pixel centre (x+0.5, y+0.5); X = z*K^-1*p; X2 = R*X + t; p2 = K*X2/X2.z; flow = p2 - p1.
 * intrinsics = (fx,fy,cx,cy) normalised by width/height (multivih5datareaderop/README.md:149-151).
 * Invalid depth (<=0 or non finite after the optional inversion) gives NaN flow.
 * gate != 0 additionally applies blocks_original.py:163-168: flow = (|flow|_2 < 1) ? flow : 0
 * (NaN compares false, so NaN -> 0).
 * ------------------------------------------------------------------------------------------- */
void ref_depth_to_flow(float *out, const float *depth, const float *intrinsics, const float *rotation,
                       const float *translation, int N, int H, int W, int inverse_depth,
                       int normalize_flow, int gate)
{
    const int hw = H * W;
    for (int n = 0; n < N; ++n) {
        const float *K = intrinsics + 4 * n;
        const float fx = K[0] * W, fy = K[1] * H, cx = K[2] * W, cy = K[3] * H;
        const float ifx = 1.0f / fx, ify = 1.0f / fy;
        float R[9];
        ref_angleaxis_to_rotation(rotation + 3 * n, R);
        const float *t = translation + 3 * n;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float d = depth[(size_t)n * hw + y * W + x];
                float fxo = NAN, fyo = NAN;
                if (inverse_depth) d = 1.0f / d;
                if (d > 0.0f && isfinite(d)) {
                    const float px = x + 0.5f, py = y + 0.5f;
                    const float X = d * ((px - cx) * ifx), Y = d * ((py - cy) * ify), Z = d;
                    const float X2 = R[0] * X + R[1] * Y + R[2] * Z + t[0];
                    const float Y2 = R[3] * X + R[4] * Y + R[5] * Z + t[1];
                    const float Z2 = R[6] * X + R[7] * Y + R[8] * Z + t[2];
                    const float p2x = fx * X2 / Z2 + cx, p2y = fy * Y2 / Z2 + cy;
                    fxo = p2x - px;
                    fyo = p2y - py;
                    if (normalize_flow) { fxo /= W; fyo /= H; }
                }
                if (gate) {
                    const float nrm = sqrtf(fxo * fxo + fyo * fyo);
                    if (!(nrm < 1.0f)) { fxo = 0.0f; fyo = 0.0f; }
                }
                out[((size_t)n * 2 + 0) * hw + y * W + x] = fxo;
                out[((size_t)n * 2 + 1) * hw + y * W + x] = fyo;
            }
    }
}

/* ---------------------------------------------------------------------------------------------
 * sops.warp2d -- call sites blocks_original.py:171-176, :336, :339 (normalized=True,
 * border_mode='value', default border_value 0).  Backward bilinear warp:
 * out(x,y) = in(x + dx*W, y + dy*H); taps outside the image contribute border_value
 * (border_mode 1 = 'value') or the clamped pixel (border_mode 0 = 'clamp').
 * ------------------------------------------------------------------------------------------- */
void ref_warp2d(float *out, const float *in, const float *disp, int N, int C, int H, int W,
                int normalized, int border_mode, float border_value)
{
    const int hw = H * W;
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float dx = disp[((size_t)n * 2 + 0) * hw + y * W + x];
                float dy = disp[((size_t)n * 2 + 1) * hw + y * W + x];
                if (normalized) { dx *= W; dy *= H; }
                const float sx = x + dx, sy = y + dy;
                const float fx0 = floorf(sx), fy0 = floorf(sy);
                const float a = sx - fx0, b = sy - fy0;
                /* keep the integer conversion defined for huge / non finite displacements */
                const int finite = isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
                const int x0 = finite ? (int)fx0 : -2, y0 = finite ? (int)fy0 : -2;
                for (int c = 0; c < C; ++c) {
                    const float *p = in + ((size_t)n * C + c) * hw;
                    float v[4];
                    for (int k = 0; k < 4; ++k) {
                        int xi = x0 + (k & 1), yi = y0 + (k >> 1);
                        if (border_mode == 1) {
                            v[k] = (finite && xi >= 0 && xi < W && yi >= 0 && yi < H) ? p[yi * W + xi] : border_value;
                        } else {
                            xi = xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
                            yi = yi < 0 ? 0 : (yi >= H ? H - 1 : yi);
                            v[k] = p[yi * W + xi];
                        }
                    }
                    float r;
                    if (finite)
                        r = (1.0f - a) * (1.0f - b) * v[0] + a * (1.0f - b) * v[1] + (1.0f - a) * b * v[2] + a * b * v[3];
                    else
                        r = (border_mode == 1) ? border_value : NAN;
                    out[((size_t)n * C + c) * hw + y * W + x] = r;
                }
            }
}

/* ---------------------------------------------------------------------------------------------
 * One-sided (Hestenes) Jacobi SVD of a 4x4 matrix, fixed 8 sweeps, float32.  Returns in X the
 * right singular vector of the smallest singular value.  The HIP kernel runs the very same
 * sequence of operations so the two agree to rounding.
 * ------------------------------------------------------------------------------------------- */
static void jacobi_null4(float A[4][4], float X[4])
{
    float V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 8; ++sweep)
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                float alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 4; ++i) {
                    alpha += A[i][p] * A[i][p];
                    beta += A[i][q] * A[i][q];
                    gamma += A[i][p] * A[i][q];
                }
                if (fabsf(gamma) <= 1e-30f || !(fabsf(gamma) > 1e-12f * sqrtf(alpha * beta))) continue;
                const float zeta = (beta - alpha) / (2.0f * gamma);
                const float tt = (zeta >= 0 ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                const float c = 1.0f / sqrtf(1.0f + tt * tt), s = c * tt;
                for (int i = 0; i < 4; ++i) {
                    const float ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const float vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
    int best = 0;
    float bestn = INFINITY;
    for (int j = 0; j < 4; ++j) {
        float nn = 0;
        for (int i = 0; i < 4; ++i) nn += A[i][j] * A[i][j];
        if (nn < bestn) { bestn = nn; best = j; }
    }
    for (int i = 0; i < 4; ++i) X[i] = V[i][best];
}

/* ---------------------------------------------------------------------------------------------
 * sops.flow_to_depth  (method 0) -- call sites blocks_original.py:344-351 / :353-360
 * (normalized_flow=True, inverse_depth=True): linear two-view triangulation of (p1, p1+flow) with
 * P1 = K[I|0], P2 = K[R|t], homogeneous least squares (DLT) solved by SVD; returns z in the
 * camera-1 frame (or 1/z).
 * sops.flow_to_depth2 (method 1) -- call site v2/blocks.py:362-378: closed-form least squares of
 * the depth along the camera-1 ray from the two reprojection equations in image 2.
 * The upstream source is not vendored: both are restatements of the published algorithm
 *  and are flagged "parity unpinned".
 * ------------------------------------------------------------------------------------------- */
void ref_flow_to_depth(float *out, const float *flow, const float *intrinsics, const float *rotation,
                       const float *translation, int N, int H, int W, int inverse_depth,
                       int normalized_flow, int method)
{
    const int hw = H * W;
    for (int n = 0; n < N; ++n) {
        const float *K = intrinsics + 4 * n;
        const float fx = K[0] * W, fy = K[1] * H, cx = K[2] * W, cy = K[3] * H;
        float R[9];
        ref_angleaxis_to_rotation(rotation + 3 * n, R);
        const float *t = translation + 3 * n;
        /* P2 = K [R|t] */
        float P2[3][4];
        for (int j = 0; j < 3; ++j) {
            P2[0][j] = fx * R[0 + j] + cx * R[6 + j];
            P2[1][j] = fy * R[3 + j] + cy * R[6 + j];
            P2[2][j] = R[6 + j];
        }
        P2[0][3] = fx * t[0] + cx * t[2];
        P2[1][3] = fy * t[1] + cy * t[2];
        P2[2][3] = t[2];
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float u = flow[((size_t)n * 2 + 0) * hw + y * W + x];
                float v = flow[((size_t)n * 2 + 1) * hw + y * W + x];
                if (normalized_flow) { u *= W; v *= H; }
                const float p1x = x + 0.5f, p1y = y + 0.5f;
                const float p2x = p1x + u, p2y = p1y + v;
                float z;
                if (method == 0) {
                    float A[4][4];
                    /* p1.x*P1[2] - P1[0], p1.y*P1[2] - P1[1] with P1 = K[I|0] */
                    A[0][0] = -fx; A[0][1] = 0;   A[0][2] = p1x - cx; A[0][3] = 0;
                    A[1][0] = 0;   A[1][1] = -fy; A[1][2] = p1y - cy; A[1][3] = 0;
                    for (int j = 0; j < 4; ++j) {
                        A[2][j] = p2x * P2[2][j] - P2[0][j];
                        A[3][j] = p2y * P2[2][j] - P2[1][j];
                    }
                    float X[4];
                    jacobi_null4(A, X);
                    z = X[2] / X[3];
                } else {
                    /* ray r = K^-1 p1 (z = 1); X2 = d*R*r + t; solve the two equations
                       (p2 - c) * X2.z = f * X2.xy for d in the least squares sense */
                    const float rx = (p1x - cx) / fx, ry = (p1y - cy) / fy;
                    const float qx = R[0] * rx + R[1] * ry + R[2];
                    const float qy = R[3] * rx + R[4] * ry + R[5];
                    const float qz = R[6] * rx + R[7] * ry + R[8];
                    const float ax = fx * qx - (p2x - cx) * qz, bx = (p2x - cx) * t[2] - fx * t[0];
                    const float ay = fy * qy - (p2y - cy) * qz, by = (p2y - cy) * t[2] - fy * t[1];
                    z = (ax * bx + ay * by) / (ax * ax + ay * ay);
                }
                out[(size_t)n * hw + y * W + x] = inverse_depth ? 1.0f / z : z;
            }
    }
}

/* sops.leaky_relu -- helpers.py:60-63: y = x >= 0 ? x : leak*x  (== max(leak*x, x) for 0<leak<1) */
void ref_leaky_relu(float *out, const float *in, size_t count, float leak)
{
    for (size_t i = 0; i < count; ++i) out[i] = in[i] >= 0.0f ? in[i] : leak * in[i];
}

/* sops.replace_nonfinite -- v2/losses.py:49: isfinite(x) ? x : value */
void ref_replace_nonfinite(float *out, const float *in, size_t count, float value)
{
    for (size_t i = 0; i < count; ++i) out[i] = isfinite(in[i]) ? in[i] : value;
}

/* ---------------------------------------------------------------------------------------------
 * sops.scale_invariant_gradient -- v2/losses.py:76-79 (one delta per call there, results concatenated
 * on axis 1 by the caller; slices of two channels are compared by scale_invariant_gradient_loss :82-104).
 * The op does not distinguish channels from batch: in [N,C,H,W] -> out [N*C, 2, H, W], channel 0 = x, 1 = y,
 * and the deltas of ONE call are summed with their weights (lmbspecialops documentation):
 *   gx = sum_k w_k*(u(x+d_k,y)-u(x,y))/(|u(x+d_k,y)|+|u(x,y)|+eps), a term is zero where the neighbour is
 *   outside the image; gy alike.
 * Unpinned (lmbspecialops source absent): the border rule.
 * ------------------------------------------------------------------------------------------- */
void ref_scale_invariant_gradient(float *out, const float *in, int NC, int H, int W, const int *deltas,
                                  const float *weights, int ndeltas, float epsilon)
{
    const int hw = H * W;
    for (int z = 0; z < NC; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float u = in[(size_t)z * hw + y * W + x];
                float gx = 0, gy = 0;
                for (int k = 0; k < ndeltas; ++k) {
                    const int d = deltas[k];
                    if (x + d >= 0 && x + d < W) {
                        const float un = in[(size_t)z * hw + y * W + x + d];
                        gx += weights[k] * (un - u) / (fabsf(un) + fabsf(u) + epsilon);
                    }
                    if (y + d >= 0 && y + d < H) {
                        const float un = in[(size_t)z * hw + (y + d) * W + x];
                        gy += weights[k] * (un - u) / (fabsf(un) + fabsf(u) + epsilon);
                    }
                }
                out[((size_t)z * 2 + 0) * hw + y * W + x] = gx;
                out[((size_t)z * 2 + 1) * hw + y * W + x] = gy;
            }
}

/* sops.median3x3_downsample -- examples/evaluation.py:173, v2/helpers.py:102:
 * [NC,H,W] -> [NC,ceil(H/2),ceil(W/2)], median of the 3x3 window centred at (2y,2x), clamped. */
/* total order: finite and infinite values by value, every NaN behind +inf ("NaN sorts last"), so the median of a
 * window with up to four NaNs is a number and with five or more it is NaN.  Unpinned (lmbspecialops absent). */
static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    const int nx = isnan(x), ny = isnan(y);
    if (nx || ny) return nx - ny;
    return (x > y) - (x < y);
}
void ref_median3x3_downsample(float *out, const float *in, int NC, int H, int W)
{
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    for (int z = 0; z < NC; ++z)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float v[9];
                int k = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        int yy = 2 * y + dy, xx = 2 * x + dx;
                        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                        v[k++] = in[(size_t)z * H * W + yy * W + xx];
                    }
                qsort(v, 9, sizeof(float), cmp_float);
                out[(size_t)z * Ho * Wo + y * Wo + x] = v[4];
            }
}

/* ---------------------------------------------------------------------------------------------
 * sops.depth_to_normals -- call site v2/losses.py:336-337 (inverse_depth=True), ground-truth normals
 * for the training loss.  depth [N,1,H,W], intrinsics (fx,fy,cx,cy) normalised -> normals [N,3,H,W] in
 * the camera frame.  UNPINNED / [RECALL] (lmbspecialops absent, no in-tree restatement):
 *   P(x,y) = z*((x+0.5-cx)/fx, (y+0.5-cy)/fy, 1), z = inverse_depth ? 1/d : d (pixel centre as in depth_to_flow);
 *   border pixels and pixels whose own or 4-neighbour depths are not finite and positive give NaN;
 *   per axis the one-sided difference (P - P(x-1), P(x+1) - P) with the smaller |dz| is used (keeps depth edges
 *   sharp); n = normalize(diff_y x diff_x), which points towards the camera (n = (0,0,-1) for a fronto-parallel
 *   plane).
 * ------------------------------------------------------------------------------------------- */
void ref_depth_to_normals(float *out, const float *depth, const float *intrinsics, int N, int H, int W, int inverse_depth)
{
    const int hw = H * W;
    for (int n = 0; n < N; ++n) {
        const float *K = intrinsics + 4 * n;
        const float fx = K[0] * W, fy = K[1] * H, cx = K[2] * W, cy = K[3] * H;
        const float ifx = 1.0f / fx, ify = 1.0f / fy;
        const float *D = depth + (size_t)n * hw;
        float *o = out + (size_t)n * 3 * hw;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float nx = NAN, ny = NAN, nz = NAN;
                if (x > 0 && y > 0 && x < W - 1 && y < H - 1) {
                    const int xs[5] = {x, x - 1, x + 1, x, x}, ys[5] = {y, y, y, y - 1, y + 1};
                    float P[5][3];
                    int ok = 1;
                    for (int k = 0; k < 5; ++k) {
                        float d = D[ys[k] * W + xs[k]];
                        if (inverse_depth) d = 1.0f / d;
                        if (!(d > 0.0f) || !isfinite(d)) ok = 0;
                        P[k][0] = d * ((xs[k] + 0.5f - cx) * ifx);
                        P[k][1] = d * ((ys[k] + 0.5f - cy) * ify);
                        P[k][2] = d;
                    }
                    if (ok) {
                        float dx[3], dy[3];
                        const int bx = fabsf(P[0][2] - P[1][2]) < fabsf(P[2][2] - P[0][2]);  /* 1: backward difference */
                        const int by = fabsf(P[0][2] - P[3][2]) < fabsf(P[4][2] - P[0][2]);
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
                o[0 * hw + y * W + x] = nx;
                o[1 * hw + y * W + x] = ny;
                o[2 * hw + y * W + x] = nz;
            }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Naive layer arithmetic (TF semantics) used to cross-check the fast
 * PyTorch-CPU layers in oracle/net_ref.py at small sizes.  double accumulation.
 *
 * conv: helpers.py:70-94 / :105-153 -- zero pad (ph,pw), VALID cross-correlation, stride (sh,sw),
 *       weight in TF HWIO layout [kh][kw][Cin][Cout], bias, optional leaky relu 0.1.
 * ------------------------------------------------------------------------------------------- */
void ref_conv2d_hwio(float *out, const float *in, const float *w, const float *bias, int N, int Cin, int H,
                     int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int lrelu)
{
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Cout; ++o)
            for (int y = 0; y < Ho; ++y)
                for (int x = 0; x < Wo; ++x) {
                    double acc = bias ? bias[o] : 0.0;
                    for (int a = 0; a < kh; ++a) {
                        const int iy = y * sh + a - ph;
                        if (iy < 0 || iy >= H) continue;
                        for (int b = 0; b < kw; ++b) {
                            const int ix = x * sw + b - pw;
                            if (ix < 0 || ix >= W) continue;
                            for (int i = 0; i < Cin; ++i)
                                acc += (double)in[(((size_t)n * Cin + i) * H + iy) * W + ix] *
                                       (double)w[(((size_t)a * kw + b) * Cin + i) * Cout + o];
                        }
                    }
                    float r = (float)acc;
                    if (lrelu) r = r >= 0.0f ? r : 0.1f * r;
                    out[(((size_t)n * Cout + o) * Ho + y) * Wo + x] = r;
                }
}

/* conv2d_transpose k=4 s=2 (blocks_original.py:64-75, :97-110): scatter
 * full[o,2y+a,2x+b] += in[i,y,x]*K[a][b][o][i] (TF layout [kh][kw][Cout][Cin]), + bias, activation,
 * then crop [1:2H+1, 1:2W+1] -> [N,Cout,2H,2W]. */
void ref_deconv4x4s2_crop(float *out, const float *in, const float *w, const float *bias, int N, int Cin,
                          int H, int W, int Cout, int lrelu)
{
    const int Hf = 2 * H + 2, Wf = 2 * W + 2, Ho = 2 * H, Wo = 2 * W;
    double *full = (double *)malloc(sizeof(double) * (size_t)Hf * Wf);
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Cout; ++o) {
            for (int k = 0; k < Hf * Wf; ++k) full[k] = bias ? bias[o] : 0.0;
            for (int i = 0; i < Cin; ++i)
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        const double v = in[(((size_t)n * Cin + i) * H + y) * W + x];
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b)
                                full[(2 * y + a) * Wf + 2 * x + b] +=
                                    v * (double)w[(((size_t)a * 4 + b) * Cout + o) * Cin + i];
                    }
            for (int y = 0; y < Ho; ++y)
                for (int x = 0; x < Wo; ++x) {
                    float r = (float)full[(y + 1) * Wf + x + 1];
                    if (lrelu) r = r >= 0.0f ? r : 0.1f * r;
                    out[(((size_t)n * Cout + o) * Ho + y) * Wo + x] = r;
                }
        }
    free(full);
}

/* tf.layers.dense (blocks_original.py:390-410): out = x*W + b, W [in][out] */
void ref_dense(float *out, const float *in, const float *w, const float *bias, int N, int Cin, int Cout,
               int lrelu)
{
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Cout; ++o) {
            double acc = bias ? bias[o] : 0.0;
            for (int i = 0; i < Cin; ++i) acc += (double)in[(size_t)n * Cin + i] * (double)w[(size_t)i * Cout + o];
            float r = (float)acc;
            if (lrelu) r = r >= 0.0f ? r : 0.1f * r;
            out[(size_t)n * Cout + o] = r;
        }
}

/* tf.image.resize_nearest_neighbor (blocks_original.py:475): src = min(floor(dst*in/out), in-1) */
void ref_resize_nearest(float *out, const float *in, int NC, int H, int W, int Ho, int Wo)
{
    for (int z = 0; z < NC; ++z)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                int sy = (int)floorf(y * ((float)H / Ho)), sx = (int)floorf(x * ((float)W / Wo));
                sy = sy > H - 1 ? H - 1 : sy;
                sx = sx > W - 1 ? W - 1 : sx;
                out[(size_t)z * Ho * Wo + y * Wo + x] = in[(size_t)z * H * W + sy * W + sx];
            }
}
