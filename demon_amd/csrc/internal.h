// internal.h -- shared declarations of libdemon_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>
#include <vector>

namespace demon {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: one bit per device and kernel instance (a `static` of the
// templated launcher).  Two threads racing on the same bit both set the attribute, which is harmless.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    template <class F>
    bool ensure(F &&configure)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return configure();
        if (done.load(std::memory_order_acquire) >> dev & 1ull) return true;
        if (!configure()) return false;
        done.fetch_or(1ull << dev, std::memory_order_release);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// A view of `C` channels [c0, c0+C) inside an NCHW device buffer that has `Ctot` channels per
// sample.  Writing layer outputs straight into such slices removes every tf.concat of
// blocks_original.py (:111-117, :183, :186, :364, :366, :482).
// ---------------------------------------------------------------------------------------------
struct View {
    float *base = nullptr;  // start of the underlying buffer
    int Ctot = 0, c0 = 0, C = 0, H = 0, W = 0;
    __host__ __device__ long n_stride() const { return (long)Ctot * H * W; }
    float *ptr() const { return base + (long)c0 * H * W; }
    View slice(int start, int count) const {
        View v = *this;
        v.c0 = c0 + start;
        v.C = count;
        return v;
    }
};

// One entry of the implicit-GEMM K table: where the B operand element of reduction index k comes
// from, relative to the output pixel's anchor (y*sy, x*sx) in the input plane.
// XCD-aware tile order.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own L2.  All
// workgroups of a launch's x-y plane whose linear id is congruent mod 8 therefore share an L2; this maps them onto a CONTIGUOUS run
// of the logical tile order (Cout tile fastest, then pixel tile), so the Cout tiles of one pixel tile and neighbouring pixel tiles
// (which share input halos) hit the same L2 instead of fetching the input from HBM once per XCD.  A bijection of the plane for any
// grid size, i.e. correct whatever the real dispatch order is.
__device__ __forceinline__ void xcd_tile(int mode, unsigned bx_in, unsigned by_in, unsigned gx, unsigned gy, unsigned &bx, unsigned &by)
{
    if (!mode) { bx = bx_in; by = by_in; return; }
    const unsigned T = gx * gy, L = bx_in + gx * by_in;
    const unsigned c = L & 7u, slot = L >> 3;
    const unsigned q = c * (T >> 3) + (c < (T & 7u) ? c : (T & 7u)) + slot;
    bx = q / gy;
    by = q - bx * gy;
}

// Diagnostic build only (-DDEMON_TIMELINE, tools/timeline.py): every workgroup of a contraction kernel records wall-clock
// stamps (s_memrealtime, 100 MHz) at kernel entry, after its prologue, after its K loop and after its stores have drained,
// plus the XCC / CU it ran on: record = 8 x u64 at index (linear workgroup id).  The product build compiles TlScope to nothing.
#ifdef DEMON_TIMELINE
struct TlScope {
    unsigned long long *rec;
    __device__ __forceinline__ TlScope(unsigned long long *base)
    {
        rec = nullptr;
        if (base && threadIdx.x == 0) {
            const unsigned long wg = blockIdx.x + (unsigned long)gridDim.x * (blockIdx.y + (unsigned long)gridDim.y * blockIdx.z);
            rec = base + 8 * wg;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            rec[4] = hw;
            rec[5] = xcc;
            rec[0] = wall_clock64();
        }
    }
    __device__ __forceinline__ void mark(int slot) { if (rec) rec[slot] = wall_clock64(); }
    __device__ __forceinline__ ~TlScope()
    {
        if (rec) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rec[3] = wall_clock64();
        }
    }
};
#else
struct TlScope {
    __device__ __forceinline__ TlScope(unsigned long long *) {}
    __device__ __forceinline__ void mark(int) {}
};
#endif

struct KEntry {
    int delta;  // ci*H*W + dy*W + dx   (elements)
    int dydx;   // (dy << 16) | (dx & 0xffff);  dy = -30000 marks padding rows (k >= K)
};

// Arguments of the implicit-GEMM convolution kernel (conv_mfma.hip).
//   D[co][pix] = sum_k Wp[k][co] * X[k][pix],  pix = (n, y, x) on a Hp x Wp pixel grid
// Plain / separable convs: the pixel grid is the output grid.  4x4 stride-2 transposed convs run
// as four 2x2 sub-pixel convs (gridDim.z = 4): the pixel grid is the INPUT grid and the output
// lands at (2y+py, 2x+px).
// ---- buffer resources carry the TRUE extent of what they address (round 5) ---------------------------------------------------------
// The minimal-filtering / first-layer / dense kernels read and write through raw buffer resources (uniform 64-bit base, one 32-bit
// byte offset per lane).  Until round 4 every descriptor said NUM_RECORDS = 1 GiB, so the hardware range check only ever caught the
// deliberate out-of-range marker offsets (0x7ffffff0 = "this element is zero padding / does not exist"); a real addressing bug read
// a neighbour tensor silently or page-faulted (it did once, in a test).  Now NUM_RECORDS = the bytes from the descriptor's base to
// the END of the tensor view the kernel was given (last sample, last channel plane), computed from the launch arguments with scalar
// arithmetic (a few SALU instructions per K-step, none on the vector ALU): a load past the end returns 0 and a store past the end is
// dropped BY HARDWARE.  What one range check cannot see -- the gaps between the channel slices of a concat buffer -- is covered by
// tests/test_poison_gpu.py (NaN-poisoned neighbours on every side).  Extents above 2 GiB clamp below the marker offsets.  K loops
// clamp the extent at THEIR base once and subtract a step's advance per step (one s_mul + one s_sub beside the base update):
// sound because a workgroup's valid offsets (within one sample / one weight array) plus that advance stay far below the clamp.
// A/B on one box against 1 GiB descriptors (python -m demon_amd.build --unbounded, tools/rounds/r5_call2.sh): see DESIGN.md section 5.
constexpr int kRsrcMaxBytes = 0x7fffffe0;        // < every out-of-range marker offset (0x7ffffff0, or-ed into valid offsets)
// `floats`: a 32-BIT count (every tensor of a context has fewer than 2^31 elements: create_impl refuses larger batches; the kernels'
// per-lane byte offsets are 32-bit anyway).  32-bit min / max / shift stay on the scalar ALU -- a 64-bit ordered compare does not
// exist there and would put a v_cmp_lt_i64 into the K loops, i.e. vector-ALU time the fp32 MFMAs share.
__device__ __forceinline__ int rsrc_bytes(int floats)
{
#ifdef DEMON_RSRC_UNBOUNDED   // diagnostic build (python -m demon_amd.build --unbounded): round 4's 1 GiB descriptors, for A/B timing only
    (void)floats;
    return 0x40000000;
#else
    // three scalar instructions, spelled out: left to itself the compiler clamps with v_med3_i32, the descriptor word then lives in a
    // VGPR and EVERY buffer load / store of the kernel becomes a readfirstlane "waterfall" loop
    int b;
    floats = __builtin_amdgcn_readfirstlane(floats);   // (wave-uniform by construction; an "s" operand the compiler keeps in a VGPR would not assemble)
    asm("s_max_i32 %0, %1, 0\n\ts_min_i32 %0, %0, 0x1ffffff8\n\ts_lshl_b32 %0, %0, 2" : "=s"(b) : "s"(floats) : "scc");   // < 0: nothing addressable; > 2 GiB: clamp below the markers
    return b;
#endif
}
// floats from channel plane `c` of sample `n` to the end of a [N][C] x plane view with sample stride n_stride (last plane: plane_floats long)
__device__ __forceinline__ int view_floats_left(int N, int n, long n_stride, int C, int c, long plane_stride, long plane_floats)
{
    return (N - 1 - n) * (int)n_stride + (C - 1 - c) * (int)plane_stride + (int)plane_floats;
}
// zero rows behind the transformed weights U[planes][Cin4][Mpad] of the minimal-filtering kernels: a K-step of KG groups reads 4 KG - 4
// rows past Cin4 at most (times zero inputs); shared by the allocation (demon_api.hip) and the descriptors' extents
constexpr int kWinoWeightSlackRows = 16;

struct ConvArgs {
    const float *in;    // input view base pointer (already offset to channel c0)
    float *out;         // output view base pointer (already offset to channel c0)
    const float *wp;    // packed weights [cls][Kpad][Mpad]
    const float *bias;  // [Mpad]
    const KEntry *ktab; // [cls][Kpad]
    const float *scale; // optional per-sample multiplier applied to output channel 0 (depth head), or null
    int N, H, W;        // input plane geometry
    long in_n_stride;   // elements between samples of the input buffer
    int Hp, Wp;         // pixel grid
    int sy, sx;         // anchor stride: iy0 = y*sy, ix0 = x*sx
    int Cout, Mpad, Kpad;
    int Ho, Wo;         // output plane geometry
    long out_n_stride;  // elements between samples of the output buffer
    int osy, osx;       // output placement: oy = y*osy + py, ox = x*osx + px
    int act;            // 1 = leaky relu 0.1
    long cls_w_stride;  // Kpad*Mpad
    float *ws;          // split-K workspace [cls][slice][Mpad][P]
    int ksplit;         // number of K slices (1 = fused epilogue)
    int xcd;            // 1: XCD-aware tile order (xcd_tile)
    long out_plane;     // elements between output channel planes (Ho*Wo unless the buffer is padded)
    unsigned long long *tl;  // timeline records (diagnostic build), else null
};

constexpr long kSplitKWorkspaceFloats = 16l << 20;  // 64 MiB of partial sums per stream

enum ConvTile { TILE_128x128 = 0, TILE_64x128, TILE_32x128, TILE_64x64, TILE_32x64, TILE_32x32, TILE_128x32, TILE_64x32, TILE_COUNT };
struct ConvPlan { int tile; int ksplit; };

// ksplit > 1: the K slices' partial sums go to the workspace and a conv_splitk_reduce launch follows
void launch_conv_mfma(const ConvArgs &a, ConvPlan plan, int nclasses, hipStream_t stream);
ConvPlan choose_conv_plan(int Mpad, long pixels, int nclasses, int Kpad, long ws_floats);
int conv_tile_bm(int tile);
int conv_tile_bn(int tile);

void launch_splitk_reduce(const ConvArgs &a, int nclasses, hipStream_t stream);
// demon_profile_full: when set, launch_splitk_reduce records this event on the stream in front of the reduce kernel (and sets the
// flag), so that a layer's own kernel and the reduce launch that follows it are timed separately
// compute units the launches of this thread run on: the CU count of the context's mask (demon_set_cu_mask), 0 = the whole device.  Read by the
// launchers that size a grid by the chip (tile-walking workgroups); set by the entry points that enqueue a context's work.
extern thread_local int g_active_cus;
extern thread_local hipEvent_t g_reduce_mark;
extern thread_local bool g_reduce_marked;

// ---- register-streaming contraction for the deep small-map layers (conv_stream.hip) -----------------------------------------
struct StreamArgs {
    ConvArgs c;            // geometry, pointers, epilogue and split-K fields as for the im2col kernel (c.wp / c.ktab unused)
    const float *wf;       // weights in MFMA fragment order [cls][step][Mpad/32][64][8] (stream_repack_kernel)
    const float *zero;     // >= (Cin + 1) * H * W zeros: what taps outside the image read
    long cls_wf_stride;    // floats between classes = K * Mpad
    int ntaps, csteps, nsteps;  // taps per class, Cin / 16, ntaps * csteps
    int tapdy[4][9], tapdx[4][9];  // per class and tap: input offset relative to the anchor pixel
    int dbg;               // diagnostic builds (-DDEMON_STREAM_DBG): bit 0 skip the A loads, bit 1 skip the B loads
};
constexpr int STREAM_VARIANTS = 18;  // (waves along Cout, row blocks per wave, column blocks per wave), see conv_stream.hip
int stream_variant_waves(int v);
int stream_variant_kw(int v);  // K-splitting wave groups inside a workgroup (1: none)
int stream_variant_bm(int v);
int stream_variant_bn(int v);
void launch_stream_repack(float *wf, const float *wp, int ncls, int K, int Mpad, long cls_w_stride, hipStream_t s);
void launch_conv_stream(const StreamArgs &s, int variant, int ksplit, int nclasses, hipStream_t stream);
// k x 1 + 1 x k stride-1 pair as one chained launch of a variant whose tile holds all channels of whole rows; false: not applicable
bool launch_conv_stream_chain(const StreamArgs &s1, const StreamArgs &s2, int variant, hipStream_t stream);

// ---- LDS-tiled contraction on fragment-ordered weights (conv_frag.hip; same arguments as the streaming kernel) -----------------
constexpr int FRAG_VARIANTS = 22;  // (waves along Cout, waves along pixels, row blocks per wave, column blocks per wave)
int frag_variant_bm(int v);
int frag_variant_bn(int v);
int frag_variant_kw(int v);  // K-splitting wave groups inside a workgroup (1: none)
void launch_conv_frag(const StreamArgs &s, int variant, int ksplit, int nclasses, hipStream_t stream);
bool launch_conv_frag_chain(const StreamArgs &s1, const StreamArgs &s2, int variant, hipStream_t stream);

// ---- patch-staged convolution (conv_patch.hip) ---------------------------------------------------------
constexpr int PATCH_EPT = 8;  // patch elements a thread stages per K-step
struct PatchArgs {
    const float *in;
    float *out;
    const float *wp;     // packed weights [cls][rows][Mpad], row = tap*Cin + ci (same packing as conv_mfma)
    const float *bias;
    const float *scale;
    float *ws;
    int N, Cin, H, W;
    long in_n_stride;
    int Hp, Wp;          // pixel grid per image
    int G, TH, TW, tiles_y, tiles_x;  // pixel tile = G images x TH rows x TW cols
    int sh, sw;
    int oy0[4], ox0[4];  // per class: input coordinates of patch element (0,0) relative to the tile anchor
    int PH, PW, PS;      // patch rows / cols / plane stride in LDS (floats; PS >= PH * PWL)
    int PWL;             // row pitch of the patch in LDS (>= PW): padded so that the 32 lanes of a fragment read hit 32 banks
    int tapoff[4][9];    // per class, per tap: float offset inside a plane
    int Cout, Mpad;
    long cls_w_stride;
    int Ho, Wo;
    long out_n_stride;
    int osy, osx;
    int act, ksplit, nsteps_total;
    int xcd;             // 1: XCD-aware tile order (xcd_tile)
    // magic numbers for exact unsigned division of the small prologue indices: n / d == mulhi(n, ceil(2^32 / d)) for n < 2^20, d < 2^12
    unsigned m_plane, m_pw, m_thtw, m_tw, m_tilesx, m_tilesy;
    unsigned long long *tl;  // timeline records (diagnostic build), else null
};
enum PatchTileId { PTILE_128x128 = 0, PTILE_64x128, PTILE_32x128, PTILE_64x64, PTILE_128x64, PTILE_32x64, PTILE_16x128, PTILE_DC4_32x128, PTILE_DC4_64x64, PTILE_COUNT };
bool patch_tile_is_dc4(int tile);  // fused 4-class transposed-conv kernel (deconv4_kernel)
int patch_cks(int ntaps, int tile);   // channels per K-step; 0: this tile has no kernel for ntaps
int patch_tile_mtiles(int tile, int Cout, int Mpad);  // workgroups along Cout
int patch_tile_bm(int tile);
int patch_tile_bn(int tile);
int patch_tile_threads(int tile);
size_t patch_lds_bytes(int tile, int ntaps, int G, int PS);
void launch_conv_patch(const PatchArgs &a, int tile, int ntaps, int nclasses, hipStream_t stream);

// ---- fused separable pair (conv_pair.hip): k x 1 conv + leaky relu + 1 x k conv + leaky relu in one launch ----------------------
struct PairArgs {
    const float *in;
    float *out;
    const float *w1, *b1;  // k x 1 layer: packed weights [tap*Cin + ci][Mpad1], bias [Mpad1]
    const float *w2, *b2;  // 1 x k layer: packed weights [tap*CMk + cm][Mpad2], bias [Mpad2]
    int N, Cin, H, W;
    long in_n_stride;
    int CM, CMk, CO, Mpad1, Mpad2;  // intermediate / output channels (CMk = CM: reduction channels of the second layer)
    int Hm;                // rows of the intermediate (its width is W)
    int Ho, Wo;
    long out_n_stride;
    int ph, pw;            // zeros in front of the first row (k x 1 layer) / column (1 x k layer)
    int tiles_y, tiles_x;
    int steps1, steps2;    // K-steps of the two phases
    int xcd;
};
bool conv_pair_applies(int k, int stride, int cin, int cm, int co);
void conv_pair_tiles(int Ho, int Wo, int &tiles_y, int &tiles_x);
int conv_pair_cks(int k);
bool launch_conv_pair(const PairArgs &a, int k, int stride, hipStream_t s);

// ---- minimal-filtering transposed conv (conv_wino.hip): F(2,2) x F(2,2) per sub-pixel class on v_mfma_f32_16x16x4_f32 -------------
struct WinoArgs {
    const float *in;
    float *out;
    const float *wp;     // packed weights [cls][tap*Cin + ci][Mpad] (the layout of every other kernel; transformed in registers)
    const float *bias;
    float *ws;           // split-K workspace [cls][slice][Mpad][P], partial sums in OUTPUT space
    int N, Cin, H, W;    // input geometry (= the class grid)
    long in_n_stride;
    int Cout, Mpad;
    long cls_w_stride;
    int Ho, Wo;
    long out_n_stride, out_plane;
    int act, ksplit, nsteps_total;       // K-steps of 4 input channels
    int G, TY, TX, tiles_y, tiles_x;     // workgroup tile = G images x TY x TX tiles (a tile = 2 x 2 outputs of every class)
    int PH, PW, PS;                      // patch rows / columns (2 TY + 2, 2 TX + 2), plane stride in LDS
    int xcd;
    unsigned m_plane, m_pw, m_tytx, m_tx, m_tilesx, m_tilesy;   // magic numbers for the prologue divisions
};
constexpr int WINO_VARIANTS = 7;   // tiles per workgroup: 32, 64, 48, 16 (16 output channels); 16, 32 (32 output channels); 6: 48 tiles with the reduction split in two inside the workgroup
int wino_variant_kh(int v);
int wino_variant_tn(int v);
int wino_variant_mb(int v);
bool wino_plan_geometry(WinoArgs &a, int variant, int n);
long wino_workgroups(const WinoArgs &a, int variant);
void launch_wino_deconv(const WinoArgs &a, int variant, hipStream_t stream);

// ---- 1-D minimal filtering for the k x 1 / 1 x k convs (conv_wino.hip, wino1d_tables.h) -------------------------------------------------
struct Wino1Args {
    const float *in;
    float *out;
    const float *wu;     // transformed weights U[e][Cin4][Mpad] (wino1d_repack_kernel)
    const float *bias;
    float *ws;           // split-K workspace [slice][Mpad][P], partial sums in OUTPUT space
    int N, Cin, Cin4, H, W, Ho, Wo;
    long in_n_stride;
    int Cout, Mpad;
    long out_n_stride, out_plane;
    int act, ksplit, nsteps_total;       // K-steps of 4 input channels
    int pad;                             // zeros in front of the first input sample along the filter axis
    int cross, csteps, cross_pad;        // taps across the filter axis (1, or 3: a 3 x 3 kernel as three 1 x 3 filters), channel steps per
                                         // cross tap (nsteps_total = cross * csteps), zeros in front of the first row across the axis
    int G, TY, TX, tiles_y, tiles_x;     // workgroup tile = G images x TY x TX tiles (a tile = 2 outputs along the filter axis)
    int xcd;
    unsigned m_tytx, m_tx, m_tilesx, m_tilesy;
    unsigned long long *tl;   // timeline records (diagnostic build), else null
};
constexpr int WINO1D_VARIANTS = 13;   // workgroup shapes (waves along Cout x waves along tiles x tile blocks per wave x K groups per step)
int wino1d_variant_kg(int v);
int wino1d_kind(int taps, int stride);   // -1: no minimal-filtering form built for this filter
int wino1d_nuv(int kind);
int wino1d_variant_bm(int v);
bool wino1d_variant_ok(int kind, int v);
bool wino1d_plan_geometry(Wino1Args &a, int kind, int variant, int axis, int n);
long wino1d_workgroups(const Wino1Args &a, int variant);
void launch_wino1d_repack(float *wu, const float *wp, int kind, int Cin, int Cin4, int Mpad, int cross, hipStream_t s);
bool launch_wino1d(const Wino1Args &a, int kind, int variant, int axis, hipStream_t stream);   // false: nothing was launched

// ---- 3 x 3 stride-1 convs as three 1 x 3 minimal-filtering row filters, transformed input rows stationary (conv_wino3.hip) -----------
struct Wino3Args {
    const float *in;
    float *out;
    const float *wu;     // transformed weights U[ky][e][Cin4][Mpad]: F(2,3) 4 planes per kernel row (wino1d_repack_kernel, cross = 3), F(4,3) 6 planes
    const float *bias;
    int N, Cin, Cin4, H, W;              // one zero in front of every row / column; stride 1: the output has the input's size
    int Ho, Wo, stride;                  // stride 2 (variants 16 ..): Ho = ceil(H / 2), Wo = W / 2, W a multiple of 8
    long in_n_stride;
    int Cout, Mpad;
    long out_n_stride, out_plane;
    int act, csteps;                     // K-steps of 4 KG input channels
    int rows_y, cols_x;                  // workgroup tiles per image: blocks of TN rows x blocks of 16 WN tile columns (a tile = 2 pixels of a row)
    int xcd;
    unsigned m_colsx, m_rowsy;
};
constexpr int WINO3_VARIANTS = 20;   // 8 workgroup shapes (waves along Cout x waves along columns x rows per wave x K groups per step) x {F(2,3), F(4,3)} + 4 for stride 2
int wino3_variant_bm(int v);
int wino3_variant_form(int v);  // 0: F(2,3), 1: F(4,3), 2: 3 x 3 stride 2 (polyphase F(4,2) + F(4,1))
bool wino3_variant_f4(int v);   // F(4,3) tiles of four pixels (own transformed weights: launch_wino3_repack43)
void launch_wino3_repack43(float *wu, const float *wp, int Cin, int Cin4, int Mpad, int stride, hipStream_t s);   // stride 1: 6 planes per kernel row, stride 2: 9
int wino3_variant_kg(int v);
int wino3_variant_rows(int v);
int wino3_variant_cols(int v);
bool wino3_plan_geometry(Wino3Args &a, int variant);
long wino3_workgroups(const Wino3Args &a, int variant);
bool launch_wino3(const Wino3Args &a, int variant, hipStream_t stream);   // false: nothing was launched

// ---- k x 1 / 1 x k convs with FOUR outputs per window: F(4,3) (3 taps, stride 1), F(4,3) + F(4,2) (5 taps, stride 2) (conv_wino4.hip) ----
struct Wino4Args {
    const float *in;
    float *out;
    const float *wu;     // transformed weights U[e][Cin4][Mpad] (wino4_repack_kernel)
    const float *bias;
    int N, Cin, Cin4, H, W, Ho, Wo;
    long in_n_stride;
    int Cout, Mpad;
    long out_n_stride, out_plane;
    int act, pad, csteps;                // zeros in front of the first sample along the filter axis; K-steps of 4 KG input channels
    int rows_y, cols_x;                  // workgroup tiles per image: blocks of TN lines x blocks of 16 WN positions (conv_wino4.hip)
    int sub_shift, pb_shift;             // narrow maps: 2^sub_shift lines side by side in a block, 2^pb_shift positions per line
    int xcd;
    unsigned m_colsx, m_rowsy;
    int gx, gy;                          // the tile grid: N rows_y cols_x pixel tiles x channel blocks (set by the launcher; a persistent launch has fewer workgroups)
    unsigned m_gx;                       // magic number of the division by gx (tile-walking launches split a linear tile index with it)
    // round 6, plan field ksplit = 3 ("flat"): the lines of ALL images form one sequence (line g -> image g / lines_img, line g % lines_img), so
    // that maps whose lines per image do not fill a workgroup's TN x 2^sub_shift lines (12 x 16: 12 rows / 3 tile rows against 8 or 16) leave no
    // empty slots.  rows_y then counts the line blocks of the whole batch and a unit's image enters its load / store offset.
    int flat, lines_img, nlines;         // lines per image, N lines_img
    unsigned m_lines;                    // magic number of the division by lines_img
    int in_img_bytes, out_img_bytes;     // 4 in_n_stride, 4 out_n_stride (the host refuses batches whose last image would not fit a 31-bit offset)
};
constexpr int WINO4_VARIANTS = 14;   // workgroup shapes (waves along Cout x waves along positions x lines per wave x K groups per step); 9 ..: three lines per wave
int wino4_kind(int taps, int stride);   // 0: 3 taps stride 1, 1: 5 taps stride 2, -1: none
int wino4_nuv(int kind);
int wino4_variant_bm(int v);
int wino4_variant_kg(int v);
bool wino4_variant_ok(int kind, int v);
bool wino4_plan_geometry(Wino4Args &a, int kind, int variant, int axis, bool flat = false);   // flat: false when it would not save a line block
long wino4_workgroups(const Wino4Args &a, int variant);
void launch_wino4_repack(float *wu, const float *wp, int kind, int Cin, int Cin4, int Mpad, hipStream_t s);
// persist: tile-walking workgroups (a whole number of tiles each, next tile's first loads under the current tile's epilogue); false when the
// tiles fit the chip in one round anyway (nothing to walk) or nothing was launched
int wino4_persist_grid(const Wino4Args &a, int kind, int variant);   // workgroups of the persistent launch; 0: not applicable
bool launch_wino4(const Wino4Args &a, int kind, int variant, int axis, hipStream_t stream, bool persist = false);   // false: nothing was launched

// ---- weight-streaming dense layer at small batch (dense_stream.hip): dense5 (v2), motion_fc1 -----------------------------------------------
struct DenseArgs {
    const float *x;      // activations [N][x_n_stride], K consecutive floats per sample
    float *out;          // [N][out_n_stride], Cout consecutive floats per sample
    const float *wd;     // weights re-blocked to [Mpad / 128][K][128] (dense_repack_kernel)
    const float *bias;   // [Mpad]
    float *ws;           // split-K workspace [slice][Mpad][N]
    int N, K, Cout, Mpad;
    long x_n_stride, out_n_stride;
    int act, ksplit;
};
bool dense_stream_geometry_ok(int K, int Mpad, int ksplit);
long dense_stream_workgroups(const DenseArgs &a);
constexpr int DENSE_VARIANTS = 2;   // 0: default cache policy on the weight loads, 1: non-temporal
void launch_dense_repack(float *wd, const float *wp, int K, int Mpad, hipStream_t s);
void launch_dense_stream(const DenseArgs &a, int variant, hipStream_t stream);   // + dense_reduce_kernel when ksplit > 1

// ---- the first layer of the blocks, a k x 1 conv over very few input channels with the weights in registers (conv_thin.hip) ------------
struct ThinArgs {
    const float *in;
    float *out;
    const float *wp;     // packed weights [tap * Cin + ci][Mpad]
    const float *bias;
    int N, Cin, H, W;
    long in_n_stride;
    int Cout, Mpad, Ho, Wo;
    long out_n_stride, out_plane;
    int pad, act, xcd;
    int tiles_y, tiles_x;   // (set by the launcher)
};
bool conv_thin_shape_ok(int kh, int kw, int sh, int sw, int ph, int pw, int Cin, int Mpad, int W, int Wo);
void launch_conv_thin(ThinArgs a, hipStream_t stream);

// ---- 1 x k stride-2 convs with <= 32 channels on both sides, whole reduction out of LDS (conv_row.hip) ---------------------------------
struct RowArgs {
    const float *in;
    float *out;
    const float *wu;     // transformed weights U[e][Cin4][32] (wino1d_repack_kernel)
    const float *bias;
    int N, Cin, Cin4, H, W;
    long in_n_stride;
    int Cout, Ho, Wo;
    long out_n_stride, out_plane;
    int pad, act;
    int tiles_y, tiles_x;   // (set by the launcher)
};
bool conv_row_shape_ok(int kh, int kw, int sh, int sw, int ph, int pw, int Cin, int Mpad, int W, int Wo);
bool launch_conv_row(RowArgs a, int taps, hipStream_t stream);

// ---- tiny heads (conv_small.hip): VALU direct conv for Cout <= 4, fused motion tail -------------------------------------
struct SmallConvArgs {
    const float *in;
    float *out;
    const float *wp;     // packed weights [k = tap*Cin + ci][Mpad]
    const float *bias;
    const float *scale;  // optional per-sample multiplier of channel 0
    int Cin, Cout, Mpad, H, W, act;
    long in_n_stride, out_n_stride;
};
bool conv_small_applies(int kh, int kw, int sh, int sw, int Cin, int Cout);
void launch_conv_small(const SmallConvArgs &a, int N, hipStream_t s);
void launch_motion_tail(const float *x, const float *w2, const float *b2, const float *w3, const float *b3, float *motion,
                        float *rot, float *trans, float *scale, int N, int K2, int M2pad, int M3pad, hipStream_t s);

// ---- op launchers (ops.hip) --------------------------------------------------------------------
void launch_depth_to_flow(float *out, const float *depth, long depth_n_stride, const float *intrinsics,
                          const float *rotation, const float *translation, int N, int H, int W,
                          long out_n_stride, int inverse_depth, int normalize_flow, int gate, hipStream_t s);
void launch_flow_to_depth(float *out, long out_n_stride, const float *flow, long flow_n_stride,
                          const float *intrinsics, const float *rotation, const float *translation, int N, int H,
                          int W, int inverse_depth, int normalized_flow, int method, float clip_hi, hipStream_t s);
void launch_warp2d(float *out, long out_n_stride, const float *in, long in_n_stride, const float *disp,
                   long disp_n_stride, int N, int C, int H, int W, int normalized, int border_mode,
                   float border_value, hipStream_t s);
// fused extra-input assembly of the iterative blocks and of the refinement net (one launch each; bit-identical to the op chains)
void launch_assemble_flow_inputs(float *extra, long extra_n_stride, const float *img2, long img2_n_stride, const float *dn,
                                 long dn_n_stride, const float *intrinsics, const float *rotation, const float *translation, int N, int H,
                                 int W, hipStream_t s);
void launch_assemble_dm_inputs(float *extra, long extra_n_stride, const float *img2, long img2_n_stride, const float *flowconf,
                               long fc_n_stride, const float *intrinsics, const float *rotation, const float *translation, int N, int H,
                               int W, int with_depth, int method, float clip_hi, hipStream_t s);
void launch_assemble_refine_input(float *out, long out_n_stride, const float *image, long image_n_stride, const float *depth2,
                                  long depth_n_stride, int N, int H, int W, int factor, hipStream_t s);
void launch_leaky_relu(float *out, const float *in, long count, float leak, hipStream_t s);
void launch_replace_nonfinite(float *out, const float *in, long count, float value, hipStream_t s);
void launch_sig(float *out, const float *in, int NC, int H, int W, const int *deltas, const float *weights,
                int ndeltas, float eps, hipStream_t s);
void launch_median3x3_downsample(float *out, const float *in, int NC, int H, int W, hipStream_t s);
void launch_depth_to_normals(float *out, const float *depth, const float *intrinsics, int N, int H, int W, int inverse_depth,
                             hipStream_t s);
void launch_pointwise_l2_partial(float *partial, const float *inp, const float *gt, int N, int C, int HW, float epsilon, hipStream_t s);
// copies C channels of a view into another view (same H, W)
void launch_copy_channels(float *dst, long dst_n_stride, const float *src, long src_n_stride, int N, int C,
                          long HW, hipStream_t s);
// nearest-neighbour upsample by an integer factor into a channel slice
void launch_upsample_nearest(float *dst, long dst_n_stride, const float *src, long src_n_stride, int N, int C,
                             int H, int W, int factor, hipStream_t s);
// splits the motion vector [N,7] into rotation [N,3], translation [N,3], scale [N,1]
void launch_split_motion(const float *motion, float *rot, float *trans, float *scale, int N, hipStream_t s);

}  // namespace demon
