// conv_wino.hip -- minimal-filtering (Winograd / Toom-Cook) forms of the transposed convs and of the k x 1 / 1 x k / 3 x 3 convs, on fp32 MFMA.
//
// The transposed convs (blocks_original.py:64-75, :97-110; `refine*/upconv`, 27 % of a batch-32 pass at 105-113 TFLOP/s, i.e. at
// what the matrix pipe gives under load) are four 2 x 2 sub-pixel convolutions.  F(2,2) computes two outputs of a 2-tap filter with
// 3 multiplications instead of 4:
//       o0 = d0*g_lo + d1*g_hi            m0 = (d0 - d1)*g_lo          o0 = m0 + m1
//       o1 = d1*g_lo + d2*g_hi            m1 =  d1*(g_lo + g_hi)       o1 = m1 + m2
//                                         m2 = (d2 - d1)*g_hi
// (all transform coefficients are +-1: no scaling, no cancellation beyond that of the direct sum), nested in y and x: a 2 x 2 block
// of outputs of one sub-pixel class costs 9 multiply-accumulates per input channel instead of 16 -- 0.5625 x the MFMAs of
// deconv4_kernel for the same result (to summation order).  GEMM view per class and (u,v) in {0,1,2}^2:
//       M[u][v][co][tile] = sum_ci U[u][v][ci][co] * T[u][v][ci][tile],     tile = a 2 x 2 block of class outputs
//       U[u][v] = sum of the taps G[ty][tx] with ty in S(u), tx in S(v), S(0) = {1}, S(1) = {0,1}, S(2) = {0}
//       T = the 3 x 3 input neighbourhood d of the tile transformed by rows, then columns: (d0 - d1, d1, d2 - d1)
//       O[a][b] = M[a][b] + M[a][b+1] + M[a+1][b] + M[a+1][b+1]
// Neither transform touches memory: a lane reads the 9 input values of its tile (and channel) from the staged patch and forms the 9
// T values with 12 subtractions; a wave reads the 4 taps of its class from LDS and forms the 9 U fragments with 5 additions -- so
// the weights stay in the packed layout every other kernel uses and the operand traffic per MFMA is that of deconv4_kernel.
//
// v_mfma_f32_16x16x4_f32: a wave owns ONE sub-pixel class, 16 output channels and TN blocks of 16 tiles (9 accumulators of 4
// registers per block); lane = (tile, channel of the K group).  A workgroup = the 4 classes of one 16-channel x 16*TN-tile block,
// sharing the staged input patch (union patch with origin (-1,-1), columns de-interleaved by parity so that the lanes of
// consecutive tiles read consecutive LDS words) and the [class][tap][k][16] weight tile.
#include <type_traits>

#include "internal.h"
#include "wino1d_tables.h"

namespace demon {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ int wdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }  // magic 0 <=> d == 1
}

// The LDS row pitch of the patch is compile-time (the 9 reads of a tile are then one base register + immediate offsets); TX <= 16
// tiles per row of the workgroup tile (34 patch columns).  Lanes of consecutive tiles read every second word: with a pitch of
// 8 (mod 16) words two tile rows of 8 tiles fall on different even banks, and an ODD plane stride puts the other channel of a
// 32-lane LDS access group on the odd banks -- conflict free for TX = 16 and TX = 8.
constexpr int WINO_PWL = 40, WINO_NT = 256, WINO_CKS = 4;

// MB: 16-channel blocks per wave (round 4).  The transformed B fragments of a tile block then feed MB MFMAs each, so the 12 VALU
// instructions and 9 LDS reads of the input transform are spread over twice the matrix work, and half as many workgroups stage the
// same input patch.
// KH = 2 (round 6, variant 6: the 6 x 8 maps of refine4): the reduction is split INSIDE the workgroup.  blockIdx.z is the output row parity py;
// the four waves are (K half kh, column parity px): a K-step stages 2 x 4 channels -- channels [4 s, 4 s + 4) for the waves kh = 0 and
// [Cin / 2 + 4 s, ...) for kh = 1 -- and the two halves' outputs are added through LDS in front of the stores.  256 workgroups for a batch
// of 32 (8 image groups x 16 channel blocks x 2 row parities) with the whole reduction in one launch: no partial sums in HBM, no reduce launch
// (the split-K form this replaces ran 128 workgroups x 3 .. 4 K slices and a reduce kernel).
template <int TN, int EPT, int OCC, int MB, int KH>
__global__ __launch_bounds__(WINO_NT, OCC) void wino_deconv_kernel(WinoArgs a)
{
    constexpr int NT = WINO_NT, CKS = WINO_CKS, PWL = WINO_PWL;
    constexpr int PCH = KH * CKS;           // channel planes of one image in the staged patch
    constexpr int ASZ1 = 4 * 4 * CKS * 16;  // floats of one 16-channel weight tile: [class][tap][k][16 channels]  (KH = 2: [kh][px][tap][k][16])
    constexpr int ASZ = MB * ASZ1;          // [mb][class][tap][k][16]
    static_assert(KH == 1 || MB == 1, "the in-workgroup K split is built for one channel block per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;               // [2][ASZ]
    const int patch_floats = a.G * PCH * a.PS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = KH == 2 ? wave >> 1 : 0;                                  // K half of this wave
    const int py = KH == 2 ? (int)blockIdx.z : wave >> 1, px = wave & 1;   // this wave's sub-pixel class (KH = 1: cls = wave)
    const int l15 = lane & 15, lk = lane >> 4;
    const int zs = KH == 2 ? 0 : blockIdx.z;
    const int khalf = KH == 2 ? a.nsteps_total / 2 : 0;                      // K-steps of one half (nsteps_total even: wino_plan_geometry)
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * (16 * MB);
    const int tyg = wdiv((int)bx, a.m_tilesx);
    const int tx = (int)bx - tyg * a.tiles_x;
    const int tgrp = wdiv(tyg, a.m_tilesy);
    const int ty = tyg - tgrp * a.tiles_y;
    const int n0 = tgrp * a.G;
    const int y_org = ty * a.TY * 2 - 1, x_org = tx * a.TX * 2 - 1;   // input coordinates of patch element (0,0)
    const float *__restrict__ in0 = a.in + (long)n0 * a.in_n_stride;

    // fp32 MFMA and the vector ALU share a SIMD's execution time (rocprofv3: SQ_VALU_MFMA_BUSY_CYCLES + 4 * SQ_ACTIVE_INST_VALU adds
    // up to the kernel's duration), so every VALU instruction in the K loop costs an eighth of an MFMA.  The loop therefore carries
    // no address arithmetic and no masking: operands come through buffer loads (uniform base advanced by SALU, per-thread byte
    // offsets fixed for the whole kernel, elements outside the image carry an out-of-range offset and read as zero), and every
    // LDS address of both buffers is precomputed.
    constexpr int OOB = 0x7ffffff0;        // >= num_records of the buffer resources (rsrc_bytes, internal.h): reads as 0
    // ---- patch loader: element e = tid + i*NT of the [G*CKS][PH][PW] patch, decoded once
    const int plane_elems = a.PH * a.PW;
    const int nelem = a.G * PCH * plane_elems;
    const int dummy_off = 2 * ASZ + 2 * patch_floats + tid;  // floats from smem: this thread's dummy slot (branch-free staging)
    int goff[EPT], lds_p[2][EPT];
    unsigned lastbits = 0;                 // elements whose channel exists in the LAST K-step (Cin not a multiple of 4)
    const int last_c0 = (a.nsteps_total - 1) * CKS;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        goff[i] = OOB;
        lds_p[0][i] = lds_p[1][i] = dummy_off;
        if (e < nelem) {
            const int pl = wdiv(e, a.m_plane), pos = e - pl * plane_elems;
            const int g = pl / PCH, cc = pl - g * PCH;
            const int c = cc % CKS, ch = (cc / CKS) * khalf * CKS + c;   // channel inside the K-step / relative to the step's first channel (KH = 2: the second half starts Cin / 2 further)
            const int pr = wdiv(pos, a.m_pw), pc = pos - pr * a.PW;
            const int gy = y_org + pr, gx = x_org + pc;
            const bool ok = ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W) & (n0 + g < a.N);
            if (ok) goff[i] = 4 * (g * (int)a.in_n_stride + ch * a.H * a.W + gy * a.W + gx);
            lds_p[0][i] = 2 * ASZ + pl * a.PS + pr * PWL + pc;
            lds_p[1][i] = lds_p[0][i] + patch_floats;
            lastbits |= ((last_c0 + c < a.Cin) ? 1u : 0u) << i;
        }
    }
    const bool mask_last = (a.Cin & 3) != 0;
    // ---- weight loader: one 16-byte chunk per thread: (class, tap, k, 4 channels) <-> packed row tap*Cin + c0 + k of that class
    int aoff;
    {
        const int slot = tid >> 6, tap = (tid >> 4) & 3, k = (tid >> 2) & 3, c4 = tid & 3;
        const int cls = KH == 2 ? 2 * py + (slot & 1) : slot;          // (KH = 2: slot = (K half, column parity) like the waves)
        const int k0 = KH == 2 ? (slot >> 1) * khalf * CKS : 0;
        aoff = 4 * (int)((long)cls * a.cls_w_stride + (long)(tap * a.Cin + k0 + k) * a.Mpad + m0 + c4 * 4);
    }

    // ---- fragment addressing: lane = (tile l15 of block tb, channel lk); float index of d[0][0] of the tile in either patch buffer
    const int ntile = a.G * a.TY * a.TX;
    int rb[2][TN], ra[2];
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        int q = tb * 16 + l15;
        if (q >= ntile) q = 0;   // padding lanes read a valid address; masked at the store
        const int g = wdiv(q, a.m_tytx), rem = q - g * (a.TY * a.TX);
        const int r = wdiv(rem, a.m_tx), c = rem - r * a.TX;
        rb[0][tb] = 2 * ASZ + (g * PCH + kh * CKS + lk) * a.PS + (2 * r + py) * PWL + 2 * c + px;
        rb[1][tb] = rb[0][tb] + patch_floats;
    }
    ra[0] = wave * (4 * CKS * 16) + lane;
    ra[1] = ra[0] + ASZ;
    // keep the second buffer's addresses in registers of their own: re-deriving them costs a VALU add per LDS access in the K loop
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) asm volatile("" : "+v"(rb[1][tb]));
    asm volatile("" : "+v"(ra[1]));
#pragma unroll
    for (int i = 0; i < EPT; ++i) asm volatile("" : "+v"(lds_p[1][i]));

    // extents of the two operand streams from THIS workgroup's base (clamped once, internal.h: rsrc_bytes); a K-step only subtracts its advance
    const int in_bytes0 = rsrc_bytes(view_floats_left(a.N, n0, a.in_n_stride, a.Cin, 0, (long)a.H * a.W, (long)a.H * a.W));
    const int w_bytes0 = rsrc_bytes(4 * (int)a.cls_w_stride);   // [4 classes][Krows][Mpad]

    floatx4 acc[MB][TN][9];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int tb = 0; tb < TN; ++tb)
#pragma unroll
            for (int q = 0; q < 9; ++q) acc[mb][tb][q] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    float pregA[EPT], pregB[EPT];
    floatx4 aregA[MB], aregB[MB];
    auto load_tiles = [&](float (&preg)[EPT], floatx4 (&areg)[MB], int step) {
        const auto prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in0 + (long)step * CKS * a.H * a.W), 0, in_bytes0 - 4 * step * CKS * a.H * a.W, 0x00020000);
        const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp + (long)step * CKS * a.Mpad), 0, w_bytes0 - 4 * step * CKS * a.Mpad, 0x00020000);
        if (mask_last && step == a.nsteps_total - 1) {   // (uniform) channels past Cin read as zero
#pragma unroll
            for (int i = 0; i < EPT; ++i)
                preg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, ((lastbits >> i) & 1u) ? goff[i] : OOB, 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) preg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, goff[i], 0, 0));
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) areg[mb] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff + 64 * mb, 0, 0));
    };
    auto store_tiles = [&](const float (&preg)[EPT], const floatx4 (&areg)[MB], int buf) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) smem[lds_p[buf][i]] = preg[i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) *reinterpret_cast<floatx4 *>(As + buf * ASZ + mb * ASZ1 + tid * 4) = areg[mb];
    };
    // ---- K loop.  One K-step = 4 input channels = one MFMA K group; global loads run two K-steps ahead (two register sets).
    // The one barrier of a step sits in front of the LAST tile block's MFMAs:
    //   blocks 0 .. TN-2 : MFMAs of block tb | LDS reads of block tb+1 | the set holding step s+1 -> the other LDS buffer | loads of step s+3
    //   barrier          : step s+1 is complete in the other buffer
    //   block TN-1       : its MFMAs | LDS reads of the taps and of block 0 of step s+1
    // so no MFMA waits for an LDS read issued just before it, and nothing but the barrier itself stops the matrix pipe.
    // Buffer safety: the buffer written during step s held step s-1, whose last reads were issued before the barrier of step s-1.
    const int per_slice = KH == 2 ? khalf : (a.nsteps_total + a.ksplit - 1) / a.ksplit;
    const int s_begin = zs * per_slice;
    const int nsteps = KH == 2 ? khalf : min(a.nsteps_total, s_begin + per_slice) - s_begin;
    auto phys = [&](int x) { return s_begin + min(x, nsteps - 1); };  // a run-ahead past the end of the slice re-reads its last step
    if (nsteps > 0) {
        load_tiles(pregA, aregA, phys(0));
        load_tiles(pregB, aregB, phys(1));
        store_tiles(pregA, aregA, 0);
        load_tiles(pregA, aregA, phys(2));
    }
    __syncthreads();
    float U[MB][9], gn[MB][4], d[9];
    auto read_taps = [&](int buf) {
        const float *A = smem + ra[buf];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {   // taps (ty,tx) = (0,0), (0,1), (1,0), (1,1)
            gn[mb][0] = A[mb * ASZ1 + 0]; gn[mb][1] = A[mb * ASZ1 + 64]; gn[mb][2] = A[mb * ASZ1 + 128]; gn[mb][3] = A[mb * ASZ1 + 192];
        }
    };
    auto make_u = [&]() {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const float *g = gn[mb];
            const float s23 = g[2] + g[3], s13 = g[1] + g[3], s02 = g[0] + g[2], s01 = g[0] + g[1];
            U[mb][0] = g[3]; U[mb][1] = s23;       U[mb][2] = g[2];
            U[mb][3] = s13;  U[mb][4] = s01 + s23; U[mb][5] = s02;
            U[mb][6] = g[1]; U[mb][7] = s01;       U[mb][8] = g[0];
        }
    };
    auto read_d = [&](int buf, int tb) {
        const float *p = smem + rb[buf][tb];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            d[j * 3 + 0] = p[j * PWL + 0];
            d[j * 3 + 1] = p[j * PWL + 1];
            d[j * 3 + 2] = p[j * PWL + 2];
        }
    };
    float b[9];
    auto transform = [&]() {
        float r0[3], r2[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { r0[i] = d[i] - d[3 + i]; r2[i] = d[6 + i] - d[3 + i]; }
        b[0] = r0[0] - r0[1]; b[1] = r0[1]; b[2] = r0[2] - r0[1];
        b[3] = d[3] - d[4];   b[4] = d[4];  b[5] = d[5] - d[4];
        b[6] = r2[0] - r2[1]; b[7] = r2[1]; b[8] = r2[2] - r2[1];
    };
    auto mfmas = [&](int tb) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb][tb][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[mb][q], b[q], acc[mb][tb][q], 0, 0, 0);
    };
    auto head = [&](int buf) {   // blocks 0 .. TN-2, and the transform of the last block
#pragma unroll
        for (int tb = 0; tb + 1 < TN; ++tb) { transform(); read_d(buf, tb + 1); mfmas(tb); }
        transform();
    };
    if (nsteps > 0) {
        int s = 0;
        read_taps(0); make_u(); read_d(0, 0);
        for (; s + 2 < nsteps; s += 2) {
            head(0);
            store_tiles(pregB, aregB, 1);
            load_tiles(pregB, aregB, phys(s + 3));
            __syncthreads();
            read_taps(1); read_d(1, 0);
            mfmas(TN - 1);
            make_u();
            head(1);
            store_tiles(pregA, aregA, 0);
            load_tiles(pregA, aregA, phys(s + 4));
            __syncthreads();
            read_taps(0); read_d(0, 0);
            mfmas(TN - 1);
            make_u();
        }
        if (s + 1 < nsteps) {
            head(0);
            store_tiles(pregB, aregB, 1);
            __syncthreads();
            read_taps(1); read_d(1, 0);
            mfmas(TN - 1);
            make_u();
            ++s;
        }
        head(s & 1);
        mfmas(TN - 1);
    }

    // ---- epilogue: O[a][b] = M[a][b] + M[a][b+1] + M[a+1][b] + M[a+1][b+1]; lane = tile, registers = 4 consecutive channels
    const long P = (long)a.N * a.H * a.W;
    const long plane = a.out_plane;
    // Full-row stores: the classes px = 0 / 1 of one py own the even / odd output columns of the same rows.  The two waves swap
    // half of their outputs through LDS (the tile buffers are free now) -- wave px = 0 keeps tile row a = 0 and receives the
    // partner's, wave px = 1 keeps a = 1 -- so that a lane writes the 4 consecutive pixels (px0 b0, px1 b0, px0 b1, px1 b1) of one
    // output row as 16 bytes, 16 lanes = 256 contiguous bytes (4-byte stores of every second pixel made the big maps store bound).
    const bool wide = a.ksplit == 1 && (a.W & 1) == 0;
    float *X = smem;   // [wave][tb][e][ib][64 lanes]
    if constexpr (KH == 2) {
        // the K halves meet here: the kh = 1 waves leave their nine accumulators per tile block in LDS (behind the exchange area; the tile
        // buffers are free), the kh = 0 waves of the same column parity add them and go on alone (the kh = 1 waves only keep the barriers company)
        float *Hs = smem + 4 * TN * 4 * 2 * 64;   // [px][tb][q][e][64 lanes]
        __syncthreads();   // every wave is done reading the tile buffers
        if (kh == 1) {
#pragma unroll
            for (int tb = 0; tb < TN; ++tb)
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Hs[(((px * TN + tb) * 9 + q) * 4 + e) * 64 + lane] = acc[0][tb][q][e];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int tb = 0; tb < TN; ++tb)
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0][tb][q][e] += Hs[(((px * TN + tb) * 9 + q) * 4 + e) * 64 + lane];
        }
    }
    const bool mine = KH == 1 || kh == 0;   // this wave transforms, exchanges and stores
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
    const int mc = m0 + 16 * mb;   // first output channel of this block
    if (wide) __syncthreads();   // every wave is done reading the tile buffers (mb = 0) / the previous block's exchange
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int q = tb * 16 + l15;
        const bool qv = q < ntile;
        const int qc = qv ? q : 0;
        const int g = wdiv(qc, a.m_tytx), rem = qc - g * (a.TY * a.TX);
        const int r = wdiv(rem, a.m_tx), c = rem - r * a.TX;
        const int n = n0 + g;
        const int y0 = (ty * a.TY + r) * 2, x0 = (tx * a.TX + c) * 2;   // class-grid (= input-grid) position of O[0][0]
        const bool tv = qv && n < a.N && y0 < a.H && x0 < a.W;
        if (wide) {
            if (!mine) continue;
            float keep[4][2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float o[2][2];
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
                        o[ia][ib] = (acc[mb][tb][ia * 3 + ib][e] + acc[mb][tb][ia * 3 + ib + 1][e]) + (acc[mb][tb][(ia + 1) * 3 + ib][e] + acc[mb][tb][(ia + 1) * 3 + ib + 1][e]);
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    keep[e][ib] = px ? o[1][ib] : o[0][ib];
                    X[(((wave * TN + tb) * 4 + e) * 2 + ib) * 64 + lane] = px ? o[0][ib] : o[1][ib];
                }
            }
            // stash the kept values in the accumulator registers this block no longer needs
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[mb][tb][0][e] = keep[e][0]; acc[mb][tb][1][e] = keep[e][1]; }
            continue;
        }
        if (!tv || !mine) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = mc + 4 * lk + e;
            float o[2][2];
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
                    o[ia][ib] = (acc[mb][tb][ia * 3 + ib][e] + acc[mb][tb][ia * 3 + ib + 1][e]) + (acc[mb][tb][(ia + 1) * 3 + ib][e] + acc[mb][tb][(ia + 1) * 3 + ib + 1][e]);
            if (a.ksplit > 1) {   // partial sums in output space, layout [cls][slice][Mpad][P] (conv_splitk_reduce finishes)
                float *__restrict__ ws = a.ws + (((long)(2 * py + px) * a.ksplit + zs) * a.Mpad + co) * P + ((long)n * a.H + y0) * a.W + x0;
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
                        if (y0 + ia < a.H && x0 + ib < a.W) ws[ia * a.W + ib] = o[ia][ib];
            } else if (co < a.Cout) {
                const float b = a.bias[co];
                float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)co * plane + (long)(2 * y0 + py) * a.Wo + (2 * x0 + px);
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        float v = o[ia][ib] + b;
                        if (a.act) v = v >= 0.0f ? v : 0.1f * v;
                        if (y0 + ia < a.H && x0 + ib < a.W) ob[(long)(2 * ia) * a.Wo + 2 * ib] = v;
                    }
            }
        }
    }
    if (!wide) continue;
    __syncthreads();
    if (!mine) continue;
    // stores through a buffer resource on this workgroup's corner of the output: uniform 64-bit base, one 32-bit offset per lane;
    // padding lanes, rows past the image and channels past Cout carry an out-of-range offset (dropped by the hardware)
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)n0 * a.out_n_stride + (long)mc * plane, 0, rsrc_bytes(view_floats_left(a.N, n0, a.out_n_stride, a.Cout, mc, plane, (long)a.Ho * a.Wo)), 0x00020000);
    const int plane4 = 4 * (int)plane;
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int q = tb * 16 + l15;
        const bool qv = q < ntile;
        const int qc = qv ? q : 0;
        const int g = wdiv(qc, a.m_tytx), rem = qc - g * (a.TY * a.TX);
        const int r = wdiv(rem, a.m_tx), c = rem - r * a.TX;
        const int ya = (ty * a.TY + r) * 2 + px, x0 = (tx * a.TX + c) * 2;   // this wave's class-grid row (a = px), first column
        const bool tv = qv && n0 + g < a.N && ya < a.H && x0 < a.W;
        const int toff = tv ? 4 * (g * (int)a.out_n_stride + (2 * ya + py) * a.Wo + 2 * x0) + 4 * lk * plane4 : 0x7ffffff0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = a.bias[mc + 4 * lk + e];   // (padded to Mpad)
            const float p0 = X[((((wave ^ 1) * TN + tb) * 4 + e) * 2 + 0) * 64 + lane], p1 = X[((((wave ^ 1) * TN + tb) * 4 + e) * 2 + 1) * 64 + lane];
            const float m0v = acc[mb][tb][0][e], m1v = acc[mb][tb][1][e];
            floatx4 v = px ? floatx4{p0, m0v, p1, m1v} : floatx4{m0v, p0, m1v, p1};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] += b;
                if (a.act) v[i] = fmaxf(v[i], 0.1f * v[i]);   // == (v >= 0 ? v : 0.1 v)
            }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, (tv && mc + 4 * lk + e < a.Cout) ? toff + e * plane4 : 0x7ffffff0, 0, 0);
        }
    }
    }
}

// =====================================================================================================================================
// k x 1 / 1 x k convolutions (the separable pairs of helpers.py:105-153): two consecutive outputs along the filter axis from one
// window of inputs with fewer multiplications -- F(2,3) for the 3-tap stride-1 layers (4 instead of 6), and for the stride-2 layers
// the polyphase split (even / odd input samples see the even / odd taps as stride-1 filters) with F(2,re) + F(2,ro): 7 instead of 10
// (5 taps), 9 instead of 14 (7 taps), 11 instead of 18 (9 taps).  Transforms: wino1d_tables.h (generated, checked in exact
// rationals).  The input transform costs vector-ALU time, which fp32 MFMAs share with the matrix work, so it is NOT done per wave:
// the threads of a workgroup transform every (tile, channel) window ONCE at staging time
// and write the NUV values to LDS ([e][k][tile]); WM waves owning different 16-channel blocks read them as the MFMA B operand.
//   AXIS 0: k x 1 filter, tile = outputs (2r, c), (2r+1, c);  AXIS 1: 1 x k filter, tile = outputs (r, 2c), (r, 2c+1)
// KG: MFMA K groups (of 4 input channels) per K-step, i.e. per barrier.  MASK: Cin is not a multiple of 4 KG -- the channels of the
// last K-step that do not exist are replaced by zeros (costs NUV selects per step)
template <int KIND, int AXIS, int WM, int WN, int TN, int KG, bool MASK>
__global__ __launch_bounds__(64 * WM * WN, (Wino1D<KIND>::NUV * TN * 4 <= (KG == 1 ? 64 : (KG == 2 ? 32 : 0)) && WM * WN <= 4) ? 3 : 2) void wino1d_kernel(Wino1Args a)
{
    using K = Wino1D<KIND>;
    constexpr int NUV = K::NUV, WIN = K::WIN, STRIDE = K::STRIDE;
    constexpr int NT = 64 * WM * WN, CKS = 4 * KG;
    // filters along x: the window of tile c starts at x = 2 STRIDE c - pad, so it lies inside the three vectors of VW = 2 STRIDE pixels
    // [2 STRIDE (c-1), 2 STRIDE (c+2)): three fully coalesced 8- / 16-byte buffer loads per unit instead of WIN 4-byte loads 8 / 16
    // bytes apart (needs W % VW == 0: a vector is then entirely inside or entirely outside its image row)
    constexpr int VW = AXIS == 1 ? 2 * STRIDE : 1;
    constexpr bool VEC = VW > 1;
    constexpr int NLD = VEC ? 3 : WIN, PWD = VEC ? 3 * VW : WIN;
    constexpr int BM = 16 * WM, NTILE = 16 * TN * WN;
    constexpr int UNITS = KG * TN / WM;                // staging units (tile, channel) per thread: CKS * NTILE / NT
    constexpr int TP = NTILE + ((NTILE & 31) ? 0 : 16);   // row pitch of T: the k = 0 / 1 halves of a 32-lane LDS access on different banks
    constexpr int ASZ = NUV * CKS * BM, TSZ = NUV * CKS * TP;
    constexpr int A4 = NUV * BM * KG;                  // 16-byte chunks of the weight tile
    constexpr int APER = (A4 + NT - 1) / NT;
    static_assert((KG * TN) % WM == 0 && UNITS >= 1, "bad shape");
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // As[2][ASZ], Ts[2][TSZ], one dummy 16-byte slot per thread
    TlScope tl(a.tl);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lk = lane >> 4;
    const int zs = blockIdx.z;
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const int m0 = by * BM;
    const int tyg = wdiv((int)bx, a.m_tilesx);
    const int tx = (int)bx - tyg * a.tiles_x;
    const int tgrp = wdiv(tyg, a.m_tilesy);
    const int ty = tyg - tgrp * a.tiles_y;
    const int n0 = tgrp * a.G;
    const float *__restrict__ in0 = a.in + (long)n0 * a.in_n_stride;
    const int ntile = a.G * a.TY * a.TX;
    const int HW = a.H * a.W;

    // ---- staging units: unit i of this thread = (tile q, channel k).  a.cross > 1 (3 x 3 layers as three 1 x 3 filters whose products
    // add up in the same accumulators): the K-steps run over (cross tap ky, channel step); the load offsets are set per ky
    int goff[UNITS][NLD], tw[2][UNITS], ur[UNITS], uc[UNITS], ub[UNITS];
    unsigned lastmask = 0;
    const int last_c0 = (a.csteps - 1) * CKS;
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
        const int w = tid + i * NT;
        const int k = w / NTILE, q = w - k * NTILE;
        const bool qv = q < ntile;
        const int qc = qv ? q : 0;
        const int g = wdiv(qc, a.m_tytx), rem = qc - g * (a.TY * a.TX);
        ur[i] = ty * a.TY + wdiv(rem, a.m_tx);                       // tile-grid coordinates
        uc[i] = tx * a.TX + (rem - wdiv(rem, a.m_tx) * a.TX);
        ub[i] = (qv && n0 + g < a.N) ? 4 * (g * (int)a.in_n_stride + k * HW) : -1;   // byte offset of the unit's channel plane, -1: no such tile
        tw[0][i] = 2 * ASZ + k * TP + q;
        tw[1][i] = tw[0][i] + TSZ;
        asm volatile("" : "+v"(tw[1][i]));
        lastmask |= ((last_c0 + k < a.Cin) ? 1u : 0u) << i;
    }
    auto set_offsets = [&](int ky) {
#pragma unroll
        for (int i = 0; i < UNITS; ++i)
#pragma unroll
            for (int e = 0; e < NLD; ++e) {
                const int gy = AXIS == 0 ? 2 * STRIDE * ur[i] - a.pad + e : ur[i] + ky - a.cross_pad;
                const int gx = AXIS == 0 ? uc[i] + ky - a.cross_pad : (VEC ? VW * (uc[i] - 1 + e) : 2 * STRIDE * uc[i] - a.pad + e);
                const bool ok = (ub[i] >= 0) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
                goff[i][e] = ok ? ub[i] + 4 * (gy * a.W + gx) : OOB;
            }
    };
    set_offsets(0);
    int cur_ky = 0;
    // ---- weight loader: chunk f of the [e][channel block][k][16] tile <-> U[e][c0 + k][m0 + 16 blk + 4 c4 ..]
    int aoff[APER], aw[2][APER];
#pragma unroll
    for (int i = 0; i < APER; ++i) {
        const int f = tid + i * NT;
        const bool fv = A4 % NT == 0 || f < A4;
        const int c4 = f & 3, k = (f >> 2) % CKS, blk = (f / (4 * CKS)) % WM, e = f / (4 * CKS * WM);
        aoff[i] = fv ? 4 * (int)(((long)e * a.Cin4 + k) * a.Mpad + m0 + blk * 16 + c4 * 4) : OOB;
        aw[0][i] = fv ? f * 4 : 2 * ASZ + 2 * TSZ + tid * 4;
        aw[1][i] = fv ? f * 4 + ASZ : 2 * ASZ + 2 * TSZ + tid * 4;
    }
    int ra[2], rt[2];
    ra[0] = wm * (16 * CKS) + lane;
    ra[1] = ra[0] + ASZ;
    rt[0] = 2 * ASZ + lk * TP + wn * (16 * TN) + l15;
    rt[1] = rt[0] + TSZ;
    asm volatile("" : "+v"(ra[1]));
    asm volatile("" : "+v"(rt[1]));

    // extents of the two operand streams from THIS workgroup's base (clamped once, internal.h: rsrc_bytes); a K-step only subtracts its advance
    const int in_bytes0 = rsrc_bytes(view_floats_left(a.N, n0, a.in_n_stride, a.Cin, 0, HW, HW));
    const int w_bytes0 = rsrc_bytes((a.cross * NUV * a.Cin4 + kWinoWeightSlackRows) * a.Mpad);

    floatx4 acc[TN][NUV];
#pragma unroll
    for (int tb = 0; tb < TN; ++tb)
#pragma unroll
        for (int e = 0; e < NUV; ++e) acc[tb][e] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    // global loads run TWO K-steps ahead (two register sets): step s on buffer s&1:
    //   loads of step s+2 | MFMAs of step s | transform + LDS stores of step s+1 (loaded during step s-1) | barrier
    float pregA[UNITS][PWD], pregB[UNITS][PWD];
    floatx4 aregA[APER], aregB[APER];
    auto load_tiles = [&](float (&preg)[UNITS][PWD], floatx4 (&areg)[APER], int step) {
        int ky = 0, cs = step;
        if (a.cross > 1) {   // (uniform) steps are issued in increasing order: the offsets change twice per kernel
            ky = step / a.csteps;
            cs = step - ky * a.csteps;
            if (ky != cur_ky) { set_offsets(ky); cur_ky = ky; }
        }
        const auto prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in0 + (long)cs * CKS * HW), 0, in_bytes0 - 4 * cs * CKS * HW, 0x00020000);
        const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wu + ((long)ky * NUV * a.Cin4 + (long)cs * CKS) * a.Mpad), 0, w_bytes0 - 4 * (ky * NUV * a.Cin4 + cs * CKS) * a.Mpad, 0x00020000);
        // channels past Cin (last K-step of a cross tap, Cin not a multiple of 4 KG) are NOT read -- the planes behind the last channel
        // of the last image may lie behind the end of the allocation: their units load from the out-of-range offset, i.e. zeros
        auto units = [&](auto last_step) {   // (the masked form only in the K-steps that need it: no extra VALU in the others)
            constexpr bool LAST = decltype(last_step)::value;
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int dead = (LAST && !((lastmask >> i) & 1u)) ? OOB : 0;   // (or-ed into the offsets: OOB covers every bit of a valid offset)
#pragma unroll
                for (int e = 0; e < NLD; ++e) {
                    if constexpr (VW == 4) {
                        const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, goff[i][e] | dead, 0, 0));
#pragma unroll
                        for (int j = 0; j < 4; ++j) preg[i][4 * e + j] = v[j];
                    } else if constexpr (VW == 2) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(prsrc, goff[i][e] | dead, 0, 0));
                        preg[i][2 * e] = v[0];
                        preg[i][2 * e + 1] = v[1];
                    } else {
                        preg[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, goff[i][e] | dead, 0, 0));
                    }
                }
            }
        };
        if (MASK && cs == a.csteps - 1) units(std::true_type{});
        else units(std::false_type{});
#pragma unroll
        for (int i = 0; i < APER; ++i) areg[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i], 0, 0));
    };
    auto transform_store = [&](const float (&preg)[UNITS][PWD], const floatx4 (&areg)[APER], int buf, int step) {
        const bool last = MASK && (step % a.csteps) == a.csteps - 1;   // (uniform) channels past Cin become zeros
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            float t[NUV];
            if constexpr (VEC) {   // the window inside the 3 VW loaded values: first element at VW - pad (pad = taps / 2, or (taps - 2) / 2 for stride-2 'same')
                float d[WIN];
                if (STRIDE == 1 || a.pad == K::TAPS / 2) {
#pragma unroll
                    for (int e = 0; e < WIN; ++e) d[e] = preg[i][VW - K::TAPS / 2 + e];
                } else {
#pragma unroll
                    for (int e = 0; e < WIN; ++e) d[e] = preg[i][VW - (K::TAPS - 2) / 2 + e];
                }
                K::input(d, t);
            } else {
                K::input(preg[i], t);
            }
            if constexpr (MASK) {
                if (last) {
                    const bool dead = !((lastmask >> i) & 1u);
#pragma unroll
                    for (int e = 0; e < NUV; ++e) t[e] = dead ? 0.0f : t[e];
                }
            }
            float *T = smem + tw[buf][i];
#pragma unroll
            for (int e = 0; e < NUV; ++e) T[e * CKS * TP] = t[e];
        }
#pragma unroll
        for (int i = 0; i < APER; ++i) *reinterpret_cast<floatx4 *>(smem + aw[buf][i]) = areg[i];
    };
    // The LDS reads of a group of MFMAs are issued one group (>= 4 MFMAs = 128 cycles) before it: left to itself the compiler reads
    // two operands, waits for them, issues their two MFMAs, reads the next two ... -- with one or two waves per SIMD (the deep,
    // small maps) the matrix pipe then idles for an LDS latency after every pair.  The scheduling barriers pin LDS and MFMA
    // instructions to the written order; vector-ALU, scalar and global-memory instructions may still move across them.
    constexpr int EG = TN >= 4 ? 1 : (TN == 3 ? 2 : 4 / TN);   // (K group, e) items per MFMA group
    constexpr int NI = KG * NUV, NGRP = (NI + EG - 1) / EG;
    constexpr bool PIPE = NUV * TN * 4 + 2 * UNITS * PWD <= 150;   // (the one shape whose registers are full reads each group just in time)
    auto compute = [&](int buf) {
        const float *A = smem + ra[buf];
        const float *T = smem + rt[buf];
        float af[2][EG], tf[2][EG][TN];
        auto fetch = [&](int g, int set) {
#pragma unroll
            for (int j = 0; j < EG; ++j) {
                const int it = g * EG + j;
                if (it >= NI) continue;
                const int kg = it / NUV, e = it - kg * NUV;
                af[set][j] = A[e * (WM * 16 * CKS) + kg * 64];
#pragma unroll
                for (int tb = 0; tb < TN; ++tb) tf[set][j][tb] = T[(e * CKS + kg * 4) * TP + tb * 16];
            }
        };
        if (PIPE) fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            if (!PIPE) fetch(g, g & 1);
            else if (g + 1 < NGRP) fetch(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0x16);
#pragma unroll
            for (int j = 0; j < EG; ++j) {
                const int it = g * EG + j;
                if (it >= NI) continue;
                const int e = it % NUV;
#pragma unroll
                for (int tb = 0; tb < TN; ++tb) acc[tb][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g & 1][j], tf[g & 1][j][tb], acc[tb][e], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0x16);
        }
    };

    const int per_slice = (a.nsteps_total + a.ksplit - 1) / a.ksplit;
    const int s_begin = zs * per_slice;
    const int nsteps = min(a.nsteps_total, s_begin + per_slice) - s_begin;
    auto phys = [&](int x) { return s_begin + min(x, nsteps - 1); };   // a run-ahead past the end of the slice re-reads its last step
    if (nsteps > 0) {
        load_tiles(pregA, aregA, phys(0));
        load_tiles(pregB, aregB, phys(1));
        transform_store(pregA, aregA, 0, phys(0));
    }
    __syncthreads();
    tl.mark(1);
    {
        int s = 0;
        for (; s + 2 < nsteps; s += 2) {
            load_tiles(pregA, aregA, phys(s + 2));
            compute(0);
            transform_store(pregB, aregB, 1, phys(s + 1));
            __syncthreads();
            load_tiles(pregB, aregB, phys(s + 3));
            compute(1);
            transform_store(pregA, aregA, 0, phys(s + 2));
            __syncthreads();
        }
        if (s + 1 < nsteps) {
            compute(0);
            transform_store(pregB, aregB, 1, phys(s + 1));
            __syncthreads();
            compute(1);
        } else if (nsteps > 0) {
            compute(0);
        }
    }
    tl.mark(2);

    // ---- epilogue: the two outputs of a tile from its NUV accumulators; lane = tile, registers = 4 consecutive channels.
    // Stores go through a buffer resource on this workgroup's corner of the output (uniform 64-bit base, 32-bit offsets per lane):
    // channels past Cout, tiles past the image and padding lanes carry an out-of-range offset and are dropped by the hardware.
    const long P = (long)a.N * a.Ho * a.Wo;
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)n0 * a.out_n_stride + (long)m0 * a.out_plane, 0, rsrc_bytes(view_floats_left(a.N, n0, a.out_n_stride, a.Cout, m0, a.out_plane, (long)a.Ho * a.Wo)), 0x00020000);
    const int plane4 = 4 * (int)a.out_plane;
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int q = wn * (16 * TN) + tb * 16 + l15;
        const bool qv = q < ntile;
        const int qc = qv ? q : 0;
        const int g = wdiv(qc, a.m_tytx), rem = qc - g * (a.TY * a.TX);
        const int r = ty * a.TY + wdiv(rem, a.m_tx), c = tx * a.TX + (rem - wdiv(rem, a.m_tx) * a.TX);
        const int n = n0 + g;
        const int y0 = AXIS == 0 ? 2 * r : r, x0 = AXIS == 0 ? c : 2 * c;
        const bool tv = qv && n < a.N && y0 < a.Ho && x0 < a.Wo;
        const bool second = AXIS == 0 ? (y0 + 1 < a.Ho) : (x0 + 1 < a.Wo);
        const int step2 = AXIS == 0 ? a.Wo : 1;
        const int toff = tv ? 4 * (g * (int)a.out_n_stride + y0 * a.Wo + x0) + (wm * 16 + 4 * lk) * plane4 : OOB;
        if (a.ksplit > 1) {
            if (!tv) continue;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {   // partial sums in output space, layout [slice][Mpad][P] (conv_splitk_reduce finishes)
                const int co = m0 + wm * 16 + 4 * lk + e4;
                float m[NUV], o0, o1;
#pragma unroll
                for (int e = 0; e < NUV; ++e) m[e] = acc[tb][e][e4];
                K::output(m, o0, o1);
                float *__restrict__ ws = a.ws + ((long)zs * a.Mpad + co) * P + ((long)n * a.Ho + y0) * a.Wo + x0;
                ws[0] = o0;
                if (second) ws[step2] = o1;
            }
            continue;
        }
        // the output transform on the four channels of an accumulator register group at once (round 6): on floatx4 the compiler emits packed fp32
        // instructions (two channels each) -- half the vector instructions of the per-channel form
        floatx4 m4[NUV], p0, p1;
#pragma unroll
        for (int e = 0; e < NUV; ++e) m4[e] = acc[tb][e];
        K::output(m4, p0, p1);
        const floatx4 b4 = *reinterpret_cast<const floatx4 *>(a.bias + m0 + wm * 16 + 4 * lk);   // (bias is padded to Mpad, a multiple of 16)
        p0 += b4; p1 += b4;
        {
            const float slope = a.act ? 0.1f : 1.0f;   // leaky relu as max(v, slope v), branch-free (slope 1: the identity)
            p0 = __builtin_elementwise_max(p0, slope * p0);
            p1 = __builtin_elementwise_max(p1, slope * p1);
        }
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int col = wm * 16 + 4 * lk + e4;   // channel inside the workgroup's block
            const float v0 = p0[e4], v1 = p1[e4];
            const int off = (tv && m0 + col < a.Cout) ? toff + e4 * plane4 : OOB;
            if (AXIS == 1 && (a.Wo & 1) == 0) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v0, v1}), orsrc, off, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), orsrc, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), orsrc, (second && off != OOB) ? off + 4 * step2 : OOB, 0, 0);
            }
        }
    }
}

// U[ky][e][ci][co] = sum_t G[e][t] wp[(ky*TAPS + t)*Cin + ci][co] (cross = 1: plain k x 1 / 1 x k layers; cross = 3: the rows of a
// 3 x 3 kernel as three 1 x 3 filters); rows ci >= Cin of U stay zero
template <int KIND>
__global__ __launch_bounds__(256) void wino1d_repack_kernel(float *__restrict__ wu, const float *__restrict__ wp, int Cin, int Cin4, int Mpad, int cross)
{
    using K = Wino1D<KIND>;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Cin * Mpad) return;
    const int ci = (int)(idx / Mpad), co = (int)(idx - (long)ci * Mpad);
    for (int ky = 0; ky < cross; ++ky) {
        float w[K::TAPS];
#pragma unroll
        for (int t = 0; t < K::TAPS; ++t) w[t] = wp[((long)(ky * K::TAPS + t) * Cin + ci) * Mpad + co];
#pragma unroll
        for (int e = 0; e < K::NUV; ++e) {
            float u = 0.0f;
#pragma unroll
            for (int t = 0; t < K::TAPS; ++t) u += K::g(e, t) * w[t];
            wu[(((long)ky * K::NUV + e) * Cin4 + ci) * Mpad + co] = u;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// variants 0..3: one 16-channel block per wave, 32 / 64 / 48 / 16 tiles per workgroup; 4, 5: two blocks per wave (32 channels), 16 / 32 tiles
// 6 (round 6): 48 tiles, the reduction split in two INSIDE the workgroup, one workgroup per output row parity (wino_deconv_kernel, KH = 2)
int wino_variant_tn(int v) { return v == 0 ? 2 : (v == 1 ? 4 : (v == 2 ? 3 : (v == 5 ? 2 : (v == 6 ? 3 : 1)))); }
int wino_variant_mb(int v) { return (v == 4 || v == 5) ? 2 : 1; }
int wino_variant_kh(int v) { return v == 6 ? 2 : 1; }

size_t wino_lds_bytes(const WinoArgs &a, int tn, int mb, int kh = 1)
{
    const size_t kloop = 2ul * mb * 4 * 4 * WINO_CKS * 16 + 2ul * a.G * kh * WINO_CKS * a.PS + WINO_NT;
    const size_t exchange = 4ul * tn * 4 * 2 * 64 + (kh == 2 ? 2ul * tn * 9 * 4 * 64 : 0ul);   // epilogue: [wave][tb][e][ib][lane] (+ the second K half's accumulators)
    return sizeof(float) * (kloop > exchange ? kloop : exchange);
}

// Workgroup tile of `ntile` = 16 * TN tiles: G images x TY x TX tiles, TX <= 16.  Picks the shape that wastes the fewest tile slots.
bool wino_plan_geometry(WinoArgs &a, int variant, int n)
{
    const int ntile = 16 * wino_variant_tn(variant);
    const int kh = wino_variant_kh(variant);
    if (a.Mpad % (16 * wino_variant_mb(variant))) return false;
    // the in-workgroup K split: two equal halves of whole K-steps, full-row stores (even width), no split across workgroups on top
    if (kh == 2 && ((a.nsteps_total & 1) || (a.Cin & 3) || (a.W & 1))) return false;
    const int ity = (a.H + 1) / 2, itx = (a.W + 1) / 2;   // tiles of one image
    double best = 1e30;
    bool ok = false;
    for (int TX : {16, 8, 4, itx}) {
        if (TX > 16 || TX > itx || TX < 1) continue;
        int TY = ntile / TX;   // (tile slots beyond G * TY * TX stay empty)
        if (TY > ity) TY = ity;
        int G = 1;
        if (TY == ity && TX == itx) { G = ntile / (TY * TX); if (G < 1) G = 1; if (G > n) G = n; }
        const int tiles_y = (ity + TY - 1) / TY, tiles_x = (itx + TX - 1) / TX, groups = (n + G - 1) / G;
        const int PH = 2 * TY + 2, PW = 2 * TX + 2;
        const long elems = (long)G * kh * WINO_CKS * PH * PW;
        if ((elems + WINO_NT - 1) / WINO_NT > (kh == 2 ? 10 : 8)) continue;
        const double waste = (double)groups * tiles_y * tiles_x * ntile / ((double)n * ity * itx);   // >= 1
        const double cost = waste * (1.0 + 0.02 * (16.0 / TX));
        if (cost < best) {
            best = cost;
            ok = true;
            a.G = G; a.TY = TY; a.TX = TX; a.tiles_y = tiles_y; a.tiles_x = tiles_x; a.PH = PH; a.PW = PW;
        }
    }
    if (!ok) return false;
    a.PS = a.PH * WINO_PWL + 1;   // odd plane stride (see WINO_PWL)
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    a.m_plane = magic(a.PH * a.PW); a.m_pw = magic(a.PW); a.m_tytx = magic(a.TY * a.TX); a.m_tx = magic(a.TX);
    a.m_tilesx = magic(a.tiles_x); a.m_tilesy = magic(a.tiles_y);
    return wino_lds_bytes(a, wino_variant_tn(variant), wino_variant_mb(variant), kh) <= (kh == 2 ? 160 : 64) * 1024;
}

long wino_workgroups(const WinoArgs &a, int variant)
{
    return (long)((a.N + a.G - 1) / a.G) * a.tiles_y * a.tiles_x * (a.Mpad / (16 * wino_variant_mb(variant))) * wino_variant_kh(variant);
}

template <int TN, int OCC, int MB>
static void launch_wino_ept(const WinoArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    const long elems = (long)a.G * WINO_CKS * a.PH * a.PW;
    const int per_thread = (int)((elems + WINO_NT - 1) / WINO_NT);
    if (per_thread <= 2) hipLaunchKernelGGL((wino_deconv_kernel<TN, 2, OCC, MB, 1>), grid, dim3(WINO_NT), lds, s, a);
    else if (per_thread <= 4) hipLaunchKernelGGL((wino_deconv_kernel<TN, 4, OCC, MB, 1>), grid, dim3(WINO_NT), lds, s, a);
    else if (per_thread <= 6) hipLaunchKernelGGL((wino_deconv_kernel<TN, 6, (MB == 2 && TN == 1 ? OCC - 1 : OCC), MB, 1>), grid, dim3(WINO_NT), lds, s, a);   // (two spilled registers at 3 waves per SIMD)
    else hipLaunchKernelGGL((wino_deconv_kernel<TN, 8, (OCC > 1 && TN != 3 ? OCC - 1 : OCC), MB, 1>), grid, dim3(WINO_NT), lds, s, a);   // 8 staged elements per thread: one wave per SIMD less, no spills
}

// variant 6: the K halves inside the workgroup, grid z = output row parity; one wave per SIMD (256 workgroups of 256 threads for refine4 at batch 32)
static bool launch_wino_kh2(const WinoArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    const long elems = (long)a.G * 2 * WINO_CKS * a.PH * a.PW;
    const int per_thread = (int)((elems + WINO_NT - 1) / WINO_NT);
    static PerDeviceOnce once6, once10;
    if (per_thread <= 6) {
        if (lds > 48 * 1024 && !once6.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_deconv_kernel<3, 6, 1, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) return false;
        hipLaunchKernelGGL((wino_deconv_kernel<3, 6, 1, 1, 2>), grid, dim3(WINO_NT), lds, s, a);
    } else {
        if (lds > 48 * 1024 && !once10.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_deconv_kernel<3, 10, 1, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) return false;
        hipLaunchKernelGGL((wino_deconv_kernel<3, 10, 1, 1, 2>), grid, dim3(WINO_NT), lds, s, a);
    }
    return true;
}

void launch_wino_deconv(const WinoArgs &a, int variant, hipStream_t stream)
{
    const int groups = (a.N + a.G - 1) / a.G;
    const int tn = wino_variant_tn(variant), mb = wino_variant_mb(variant);
    if (wino_variant_kh(variant) == 2) {   // (ksplit is 1 here: run_wino clamps it for this variant)
        launch_wino_kh2(a, dim3((unsigned)(groups * a.tiles_y * a.tiles_x), (unsigned)(a.Mpad / 16), 2), wino_lds_bytes(a, tn, mb, 2), stream);
        return;
    }
    dim3 grid((unsigned)(groups * a.tiles_y * a.tiles_x), (unsigned)(a.Mpad / (16 * mb)), (unsigned)a.ksplit);
    const size_t lds = wino_lds_bytes(a, tn, mb);
    if (mb == 2) {
        if (tn == 1) launch_wino_ept<1, 3, 2>(a, grid, lds, stream);
        else launch_wino_ept<2, 2, 2>(a, grid, lds, stream);
        return;
    }
    switch (tn) {
        case 2: launch_wino_ept<2, 3, 1>(a, grid, lds, stream); break;
        case 3: launch_wino_ept<3, 2, 1>(a, grid, lds, stream); break;
        case 1: launch_wino_ept<1, 4, 1>(a, grid, lds, stream); break;
        default: launch_wino_ept<4, 2, 1>(a, grid, lds, stream); break;
    }
}


// ---- 1-D host side
// kinds: 0 = 3 taps stride 1, 1 / 2 / 3 = 5 / 7 / 9 taps stride 2.  variants: workgroup shapes (WM x WN waves, TN tile blocks per wave)
int wino1d_kind(int taps, int stride)
{
    if (taps == 3 && stride == 1) return 0;
    if (stride == 2 && (taps == 5 || taps == 7 || taps == 9)) return (taps - 3) / 2;
    return -1;
}
int wino1d_nuv(int kind) { return kind == 0 ? 4 : 5 + 2 * kind; }
struct W1Shape { int wm, wn, tn, kg; };
// 8..10: 48 / 96 tiles per workgroup -- the maps of this net are 3 * 2^k rows high (6 x 8, 12 x 16, 24 x 32 ...), so whole images fit
// a 48- or 96-tile workgroup exactly where the 64-tile shapes leave a quarter of their tile slots empty (6 x 8: 24 tiles per image)
// 11..12: ONE 16-channel block per workgroup, four waves along the tiles -- for layers with at most 16 output channels (the 64 -> 16
// conv of the refinement net's depth head at full resolution, blocks_original.py:505-511): no MFMA rows spent on channels that do not
// exist.  The input transform then has a single consumer, which only such layers have to accept.
static const W1Shape kW1Shapes[WINO1D_VARIANTS] = {{2, 2, 2, 1}, {4, 1, 4, 1}, {2, 2, 4, 1}, {4, 2, 4, 1}, {2, 2, 2, 2}, {4, 1, 4, 2}, {2, 2, 4, 2}, {4, 2, 4, 2},
                                                   {4, 1, 3, 4}, {2, 2, 3, 2}, {4, 2, 3, 4}, {1, 4, 4, 1}, {1, 4, 2, 2}};
int wino1d_variant_kg(int v) { return kW1Shapes[v].kg; }
int wino1d_variant_bm(int v) { return 16 * kW1Shapes[v].wm; }
int wino1d_variant_ntile(int v) { return 16 * kW1Shapes[v].tn * kW1Shapes[v].wn; }

static size_t wino1d_lds_bytes(int kind, int v)
{
    const int nuv = wino1d_nuv(kind), bm = wino1d_variant_bm(v), ntile = wino1d_variant_ntile(v);
    const int tp = ntile + ((ntile & 31) ? 0 : 16), nt = 64 * kW1Shapes[v].wm * kW1Shapes[v].wn, cks = 4 * wino1d_variant_kg(v);
    return sizeof(float) * (2ul * (nuv * cks * bm + nuv * cks * tp) + 4ul * nt);
}

bool wino1d_variant_ok(int kind, int v)
{
    if (kind < 0 || v < 0 || v >= WINO1D_VARIANTS) return false;
    // accumulators: NUV x TN x 4 registers per lane; two K groups per step (variants 4..7) double the staging registers: kinds 0 / 1 only
    if (wino1d_variant_kg(v) == 2 && (kind == 3 || (kind == 2 && kW1Shapes[v].tn > 2) || (kind == 1 && kW1Shapes[v].tn > 3))) return false;   // (register budget)
    if (wino1d_variant_kg(v) == 4 && kind != 0) return false;                                            // (LDS: 16 channels x NUV per step)
    return wino1d_nuv(kind) * kW1Shapes[v].tn * 4 <= 112 && wino1d_lds_bytes(kind, v) <= 160 * 1024;
}

// tile grid of one image: AXIS 0: (ceil(Ho/2), Wo), AXIS 1: (Ho, ceil(Wo/2)); workgroup tile = G images x TY x TX tiles
bool wino1d_plan_geometry(Wino1Args &a, int kind, int variant, int axis, int n)
{
    if (!wino1d_variant_ok(kind, variant) || a.Mpad % wino1d_variant_bm(variant)) return false;
    if (wino1d_variant_bm(variant) == 16 && a.Cout > 16) return false;   // (one channel block per workgroup: see kW1Shapes)
    if (kind == 1 && axis == 1 && kW1Shapes[variant].tn == 3) return false;   // (register budget: 16-byte window vectors x 3 units x 2 sets)
    if (axis == 1) {   // filters along x load their windows as 8-byte (stride 1) / 16-byte (stride 2) vectors
        const int taps = kind == 0 ? 3 : 3 + 2 * kind;
        if (kind == 0 ? ((a.W & 1) || a.pad != 1) : ((a.W & 3) || (a.pad != taps / 2 && a.pad != (taps - 2) / 2))) return false;
    }
    const int ntile = wino1d_variant_ntile(variant);
    const int gh = axis == 0 ? (a.Ho + 1) / 2 : a.Ho, gw = axis == 0 ? a.Wo : (a.Wo + 1) / 2;
    double best = 1e30;
    bool ok = false;
    for (int TX : {64, 32, 16, 8, gw}) {
        if (TX > gw || TX < 1 || TX > ntile) continue;
        int TY = ntile / TX;
        if (TY < 1) continue;
        if (TY > gh) TY = gh;
        int G = 1;
        if (TY == gh && TX == gw) { G = ntile / (TY * TX); if (G < 1) G = 1; if (G > n) G = n; }
        const int tiles_y = (gh + TY - 1) / TY, tiles_x = (gw + TX - 1) / TX, groups = (n + G - 1) / G;
        const double waste = (double)groups * tiles_y * tiles_x * ntile / ((double)n * gh * gw);
        // rows of a workgroup tile share their input windows only along the filter axis: prefer tiles long in that direction
        const double cost = waste * (1.0 + 0.01 * (axis == 0 ? TX : TY));
        if (cost < best) {
            best = cost;
            ok = true;
            a.G = G; a.TY = TY; a.TX = TX; a.tiles_y = tiles_y; a.tiles_x = tiles_x;
        }
    }
    if (!ok) return false;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    a.m_tytx = magic(a.TY * a.TX); a.m_tx = magic(a.TX); a.m_tilesx = magic(a.tiles_x); a.m_tilesy = magic(a.tiles_y);
    return true;
}

long wino1d_workgroups(const Wino1Args &a, int variant)
{
    return (long)((a.N + a.G - 1) / a.G) * a.tiles_y * a.tiles_x * ((a.Cout + wino1d_variant_bm(variant) - 1) / wino1d_variant_bm(variant));
}

void launch_wino1d_repack(float *wu, const float *wp, int kind, int Cin, int Cin4, int Mpad, int cross, hipStream_t s)
{
    const long total = (long)Cin * Mpad;
    const dim3 grid((unsigned)((total + 255) / 256));
    switch (kind) {
        case 0: hipLaunchKernelGGL(wino1d_repack_kernel<0>, grid, dim3(256), 0, s, wu, wp, Cin, Cin4, Mpad, cross); break;
        case 1: hipLaunchKernelGGL(wino1d_repack_kernel<1>, grid, dim3(256), 0, s, wu, wp, Cin, Cin4, Mpad, cross); break;
        case 2: hipLaunchKernelGGL(wino1d_repack_kernel<2>, grid, dim3(256), 0, s, wu, wp, Cin, Cin4, Mpad, cross); break;
        default: hipLaunchKernelGGL(wino1d_repack_kernel<3>, grid, dim3(256), 0, s, wu, wp, Cin, Cin4, Mpad, cross); break;
    }
}

template <int KIND, int AXIS, int WM, int WN, int TN, int KG, bool MASK>
static bool launch_w1m(const Wino1Args &a, dim3 grid, size_t lds, hipStream_t s)
{
    static PerDeviceOnce once;
    if (!once.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&wino1d_kernel<KIND, AXIS, WM, WN, TN, KG, MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) return false;
    hipLaunchKernelGGL((wino1d_kernel<KIND, AXIS, WM, WN, TN, KG, MASK>), grid, dim3(64 * WM * WN), lds, s, a);
    return true;
}

template <int KIND, int AXIS, int WM, int WN, int TN, int KG>
static bool launch_w1(const Wino1Args &a, dim3 grid, size_t lds, hipStream_t s)
{
    if constexpr (Wino1D<KIND>::NUV * TN * 4 <= 112 && (KG == 1 || KIND == 0 || (KIND == 1 && KG == 2 && TN <= 3) || (KIND == 2 && KG == 2 && TN == 2)) && !(KIND == 1 && AXIS == 1 && TN == 3)) {
        if (a.Cin % (4 * KG)) return launch_w1m<KIND, AXIS, WM, WN, TN, KG, true>(a, grid, lds, s);
        return launch_w1m<KIND, AXIS, WM, WN, TN, KG, false>(a, grid, lds, s);
    }
    return false;   // (no kernel of this shape for this filter: wino1d_variant_ok / wino1d_plan_geometry reject it first)
}

template <int KIND, int AXIS>
static bool launch_w1_variant(const Wino1Args &a, int variant, dim3 grid, size_t lds, hipStream_t s)
{
    switch (variant) {
        case 0: return launch_w1<KIND, AXIS, 2, 2, 2, 1>(a, grid, lds, s);
        case 1: return launch_w1<KIND, AXIS, 4, 1, 4, 1>(a, grid, lds, s);
        case 2: return launch_w1<KIND, AXIS, 2, 2, 4, 1>(a, grid, lds, s);
        case 3: return launch_w1<KIND, AXIS, 4, 2, 4, 1>(a, grid, lds, s);
        case 4: return launch_w1<KIND, AXIS, 2, 2, 2, 2>(a, grid, lds, s);
        case 5: return launch_w1<KIND, AXIS, 4, 1, 4, 2>(a, grid, lds, s);
        case 6: return launch_w1<KIND, AXIS, 2, 2, 4, 2>(a, grid, lds, s);
        case 7: return launch_w1<KIND, AXIS, 4, 2, 4, 2>(a, grid, lds, s);
        case 8: return launch_w1<KIND, AXIS, 4, 1, 3, 4>(a, grid, lds, s);
        case 9: return launch_w1<KIND, AXIS, 2, 2, 3, 2>(a, grid, lds, s);
        case 10: return launch_w1<KIND, AXIS, 4, 2, 3, 4>(a, grid, lds, s);
        case 11: return launch_w1<KIND, AXIS, 1, 4, 4, 1>(a, grid, lds, s);
        default: return launch_w1<KIND, AXIS, 1, 4, 2, 2>(a, grid, lds, s);
    }
}

bool launch_wino1d(const Wino1Args &a, int kind, int variant, int axis, hipStream_t stream)
{
    const int groups = (a.N + a.G - 1) / a.G;
    const int bm = wino1d_variant_bm(variant);
    dim3 grid((unsigned)(groups * a.tiles_y * a.tiles_x), (unsigned)((a.Cout + bm - 1) / bm), (unsigned)a.ksplit);   // (blocks of padding channels only would store nothing)
    const size_t lds = wino1d_lds_bytes(kind, variant);
#define W1_KIND(KK)                                                              \
    case KK:                                                                     \
        return axis == 0 ? launch_w1_variant<KK, 0>(a, variant, grid, lds, stream)  \
                         : launch_w1_variant<KK, 1>(a, variant, grid, lds, stream);
    switch (kind) {
        W1_KIND(0)
        W1_KIND(1)
        W1_KIND(2)
        W1_KIND(3)
        default: return false;
    }
#undef W1_KIND
}

}  // namespace demon
