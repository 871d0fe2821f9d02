// conv_pair.hip -- one launch for a separable pair "k x 1 conv, leaky relu, 1 x k conv, leaky relu" (helpers.py:105-153
// convrelu2_caffe_padding; v2/helpers.py:44-91 convrelu2) on the large maps of levels 1-2, where the two launches of the pair are
// memory / latency bound: the intermediate tensor (100 MB at level 1, batch 32) never leaves the chip and one pipeline fill / drain
// per pair disappears.
//
// A workgroup (4 waves) owns TH x TW = 4 x 32 output pixels of one image:
//   phase 1  k x 1 conv (stride (S,1)) for the TH x TWm intermediate pixels the tile needs (TWm = (TW-1)*S + K columns), as the
//            patch-staged MFMA GEMM of conv_patch.hip (M = intermediate channels, N = pixels, K-steps of CKS input channels x K
//            taps, input patch and weight tile double-buffered in LDS); + bias, leaky relu; columns / rows outside the image become
//            the zeros the 1 x k conv pads with; result -> LDS `mid[channel][TH][TWp]`
//   phase 2  1 x k conv (stride (1,S)) straight out of `mid` (lane = output pixel, tap = column shift), weight tiles streamed
//            through the LDS double buffer; + bias, leaky relu -> global, 128-byte rows.
// Reduction order per output: channel chunk major, tap, channel inside the chunk (both phases), like the patch kernel.
#include "internal.h"

namespace demon {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// C1 = input channels per K-step of phase 1: 2 (9 / 7 taps), 8 (3 taps), or 6 = the whole input of conv1 in ONE step (no
// double buffer, no load / barrier round trips inside phase 1: a workgroup's lifetime is a chain of memory latencies)
template <int K, int S, int MB1, int MB2, int C1>
struct PairGeom {
    static constexpr int TH = 4, TW = 32;
    static constexpr int TWm = (TW - 1) * S + K;          // intermediate columns the tile needs
    static constexpr int TWp = (TWm + 3) / 4 * 4;         // row pitch of patch and mid in LDS
    static constexpr int PH = (TH - 1) * S + K;           // input rows
    static constexpr int N1 = TH * TWp;                   // intermediate pixels (with pitch padding)
    static constexpr int NB1 = (N1 + 31) / 32;            // 32-pixel MFMA column blocks of phase 1
    static constexpr int NBW = (NB1 + 3) / 4;             // ... per wave
    static constexpr int CKS1 = C1;                       // input channels per K-step of phase 1
    static constexpr bool ONE1 = C1 == 6;                 // phase 1 is a single K-step
    static constexpr int CKS2 = K == 3 ? 8 : 2;           // intermediate channels per K-step of phase 2
    static constexpr int KD1 = K * CKS1, KD2 = K * CKS2;
    static constexpr int PPS = PH * TWp;                  // patch plane
    static constexpr int PELEMS = CKS1 * PPS;
    static constexpr int EPP = (PPS + 255) / 256;         // patch positions per thread (each for CKS1 channels)
    static constexpr int A1CH = KD1 * MB1 * 8, A1PER = (A1CH + 255) / 256;  // float4 chunks of the phase-1 weight tile
    static constexpr int A2CH = KD2 * MB2 * 8, A2PER = (A2CH + 255) / 256;
    static constexpr int MS = N1;                         // mid plane stride
    static constexpr int MID = MB1 * 32 * MS;
    static constexpr int STAGE1 = (ONE1 ? 1 : 2) * (KD1 * MB1 * 32 + PELEMS);
    // phase-2 weights: the whole [K * CM][BM2] matrix stays in LDS when it is at most 40 KB (no K-steps, no barriers in phase 2);
    // otherwise tiles of CKS2 channels stream through a double buffer
    static constexpr int W2ALL = K * MB1 * 32 * MB2 * 32;
    static constexpr bool W2RES = W2ALL * 4 <= 40 * 1024;
    static constexpr int W2CH = W2ALL / 4, W2PER = (W2CH + 255) / 256;
    static constexpr int STAGE2 = W2RES ? W2ALL : 2 * KD2 * MB2 * 32;
    static constexpr int STAGE = STAGE1 > STAGE2 ? STAGE1 : STAGE2;
    static constexpr size_t lds_bytes = sizeof(float) * (size_t)(MID + STAGE);
};

template <int K, int S, int MB1, int MB2, int C1>
__global__ __launch_bounds__(256) void conv_pair_kernel(PairArgs a)
{
    using G = PairGeom<K, S, MB1, MB2, C1>;
    constexpr int TH = G::TH, TW = G::TW, TWm = G::TWm, TWp = G::TWp, N1 = G::N1, NB1 = G::NB1, NBW = G::NBW;
    constexpr int CKS1 = G::CKS1, CKS2 = G::CKS2, KD1 = G::KD1, KD2 = G::KD2, PPS = G::PPS, PELEMS = G::PELEMS, EPP = G::EPP;
    constexpr bool ONE1 = G::ONE1;
    constexpr int A1PER = G::A1PER, A2PER = G::A2PER, MS = G::MS;
    constexpr int BM1 = MB1 * 32, BM2 = MB2 * 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *mid = smem;                   // [BM1][MS]
    float *stage = smem + G::MID;        // phase 1: As[2][KD1][BM1], Ps[2][PELEMS]; phase 2: As[2][KD2][BM2]
    float *As1 = stage, *Ps = stage + (ONE1 ? 1 : 2) * KD1 * BM1, *As2 = stage;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    unsigned bx, by;
    xcd_tile(a.xcd, blockIdx.x, 0, gridDim.x, 1, bx, by);
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = bx / per_img, trem = bx - n * per_img;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int y_org = ty * TH * S - a.ph;   // input row of patch row 0
    const int x_org = tx * TW * S - a.pw;   // intermediate (= input) column of patch / mid column 0
    const float *__restrict__ in0 = a.in + (long)n * a.in_n_stride;

    // ---- phase 1 staging: a thread owns patch positions pos = tid + i*256 of [PH][TWp], decoded once, for all CKS1 channels
    int goff[EPP];
    unsigned okbits = 0;
    const int HW = a.H * a.W;
#pragma unroll
    for (int i = 0; i < EPP; ++i) {
        const int pos = tid + i * 256;
        goff[i] = 0;
        if (pos < PPS) {
            const int py = pos / TWp, px = pos - py * TWp;
            const int gy = y_org + py, gx = x_org + px;
            const bool ok = (px < TWm) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
            if (ok) goff[i] = gy * a.W + gx;
            okbits |= (ok ? 1u : 0u) << i;
        }
    }
    int a1off[A1PER];
#pragma unroll
    for (int i = 0; i < A1PER; ++i) {
        const int q = tid + i * 256;
        const int r = q / (BM1 / 4), c4 = q - r * (BM1 / 4);
        a1off[i] = ((r / CKS1) * a.Cin + (r % CKS1)) * a.Mpad1 + c4 * 4;
    }
    float preg[CKS1][EPP];
    floatx4 areg[A1PER > A2PER ? A1PER : A2PER];
    auto load1 = [&](int step) {
        const float *__restrict__ ab = a.w1 + (long)step * CKS1 * a.Mpad1;
#pragma unroll
        for (int c = 0; c < CKS1; ++c) {
            const float *__restrict__ pb = in0 + (long)min(step * CKS1 + c, a.Cin - 1) * HW;  // channels past Cin: zeroed in store1
#pragma unroll
            for (int i = 0; i < EPP; ++i)
                if (tid + i * 256 < PPS) preg[c][i] = pb[goff[i]];
        }
#pragma unroll
        for (int i = 0; i < A1PER; ++i)
            if (tid + i * 256 < G::A1CH) areg[i] = *reinterpret_cast<const floatx4 *>(ab + a1off[i]);
    };
    auto store1 = [&](int buf, int step) {
        float *P = Ps + buf * PELEMS;
#pragma unroll
        for (int c = 0; c < CKS1; ++c) {
            const bool cok = step * CKS1 + c < a.Cin;
#pragma unroll
            for (int i = 0; i < EPP; ++i)
                if (tid + i * 256 < PPS) P[c * PPS + tid + i * 256] = (cok && ((okbits >> i) & 1u)) ? preg[c][i] : 0.0f;
        }
        float *A = As1 + buf * (KD1 * BM1);
#pragma unroll
        for (int i = 0; i < A1PER; ++i)
            if (tid + i * 256 < G::A1CH) *reinterpret_cast<floatx4 *>(A + (tid + i * 256) * 4) = areg[i];
    };

    // ---- phase 1 fragment addressing: this wave's pixel blocks wave, wave+4, wave+8 of the [TH][TWp] intermediate tile
    int pbase[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        int q = (nb * 4 + wave) * 32 + l31;
        if (q >= N1) q = 0;
        const int r = q / TWp, j = q - r * TWp;
        pbase[nb] = 4 * (r * S * TWp + j + lhi * PPS);
    }
    floatx16 acc1[MB1][NBW];
#pragma unroll
    for (int i = 0; i < MB1; ++i)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][nb][r] = 0.0f;

    load1(0);
    store1(0, 0);
    __syncthreads();
    for (int s = 0; s < a.steps1; ++s) {
        const int buf = ONE1 ? 0 : (s & 1);
        if (!ONE1 && s + 1 < a.steps1) load1(s + 1);
        const char *Pb = reinterpret_cast<const char *>(Ps + buf * PELEMS);
        const float *A = As1 + buf * (KD1 * BM1);
#pragma unroll
        for (int kk = 0; kk < KD1 / 2; ++kk) {
            const int k = 2 * kk + lhi;
            const int so = 4 * (((2 * kk) % CKS1) * PPS + ((2 * kk) / CKS1) * TWp);  // even channel of the pair, tap row
            float av[MB1];
#pragma unroll
            for (int i = 0; i < MB1; ++i) av[i] = A[k * BM1 + i * 32 + l31];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                if ((nb * 4 + wave) < NB1) {
                    const float bv = *reinterpret_cast<const float *>(Pb + pbase[nb] + so);
#pragma unroll
                    for (int i = 0; i < MB1; ++i) acc1[i][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc1[i][nb], 0, 0, 0);
                }
            }
        }
        if (!ONE1 && s + 1 < a.steps1) store1(buf ^ 1, s + 1);
        __syncthreads();
    }

    // ---- phase 2 weights
    constexpr bool W2RES = G::W2RES;
    constexpr int W2PER = G::W2PER;
    floatx4 wreg[W2RES ? W2PER : 1];
    int a2off[A2PER];
    const int w2count = K * a.CMk * (BM2 / 4);  // float4 chunks of the real matrix (rows tap*CM + cm, BM2 == Mpad2 floats each)
    if (W2RES) {
        // resident: every thread fetches its chunks now (they fly during the intermediate's epilogue) ...
#pragma unroll
        for (int i = 0; i < W2PER; ++i)
            if (tid + i * 256 < w2count) wreg[i] = *reinterpret_cast<const floatx4 *>(a.w2 + (long)(tid + i * 256) * 4);
    } else {
#pragma unroll
        for (int i = 0; i < A2PER; ++i) {
            const int q = tid + i * 256;
            const int r = q / (BM2 / 4), c4 = q - r * (BM2 / 4);
            a2off[i] = ((r / CKS2) * a.CMk + (r % CKS2)) * a.Mpad2 + c4 * 4;
        }
    }
    auto load2 = [&](int step) {
        const float *__restrict__ ab = a.w2 + (long)step * CKS2 * a.Mpad2;
#pragma unroll
        for (int i = 0; i < A2PER; ++i)
            if (tid + i * 256 < G::A2CH) areg[i] = *reinterpret_cast<const floatx4 *>(ab + a2off[i]);
    };
    auto store2 = [&](int buf) {
        float *A = As2 + buf * (KD2 * BM2);
#pragma unroll
        for (int i = 0; i < A2PER; ++i)
            if (tid + i * 256 < G::A2CH) *reinterpret_cast<floatx4 *>(A + (tid + i * 256) * 4) = areg[i];
    };
    if (!W2RES) load2(0);

    // ---- intermediate: bias, leaky relu, zero outside the image (the 1 x k conv's padding), -> mid
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int q = (nb * 4 + wave) * 32 + l31;
        if ((nb * 4 + wave) < NB1 && q < N1) {
            const int r = q / TWp, j = q - r * TWp;
            const int gxm = x_org + j, gym = ty * TH + r;
            const bool inside = (j < TWm) & ((unsigned)gxm < (unsigned)a.W) & (gym < a.Hm);
#pragma unroll
            for (int i = 0; i < MB1; ++i)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int cm = i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lhi;
                    float v = acc1[i][nb][rr] + a.b1[cm];
                    v = v >= 0.0f ? v : 0.1f * v;
                    mid[cm * MS + q] = inside ? v : 0.0f;
                }
        }
    }
    // the phase-1 staging buffers are free: every wave passed the barrier that ended the last K-step
    if (W2RES) {
#pragma unroll
        for (int i = 0; i < W2PER; ++i)
            if (tid + i * 256 < w2count) *reinterpret_cast<floatx4 *>(stage + (tid + i * 256) * 4) = wreg[i];
    } else {
        store2(0);
    }
    __syncthreads();

    // ---- phase 2: wave = output row `wave` of the tile, lane = column
    floatx16 acc2[MB2];
#pragma unroll
    for (int i = 0; i < MB2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][r] = 0.0f;
    const int mbase = 4 * (wave * TWp + l31 * S + lhi * MS);
    if (W2RES) {
        // tap major, channel pairs inside: k2 = tap*CM + cm, no barrier until the end
        const char *Mb = reinterpret_cast<const char *>(mid) + mbase;
        const float *A = stage + l31;
        for (int b = 0; b < K; ++b) {
            const float *Ab = A + (long)(b * a.CMk + lhi) * BM2;
            const char *Mt = Mb + 4 * b;
            for (int cm0 = 0; cm0 < a.CMk; cm0 += 8) {  // CM is a multiple of 8 (conv_pair_applies)
                float av[4][MB2], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cm = cm0 + 2 * u;
                    bv[u] = *reinterpret_cast<const float *>(Mt + 4 * cm * MS);
#pragma unroll
                    for (int i = 0; i < MB2; ++i) av[u][i] = Ab[cm * BM2 + i * 32];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < MB2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u], acc2[i], 0, 0, 0);
            }
        }
    } else {
        for (int s = 0; s < a.steps2; ++s) {
            const int buf = s & 1;
            if (s + 1 < a.steps2) load2(s + 1);
            const char *Mb = reinterpret_cast<const char *>(mid + (long)s * CKS2 * MS) + mbase;
            const float *A = As2 + buf * (KD2 * BM2);
#pragma unroll
            for (int kk = 0; kk < KD2 / 2; ++kk) {
                const int k = 2 * kk + lhi;
                const float bv = *reinterpret_cast<const float *>(Mb + 4 * (((2 * kk) % CKS2) * MS + (2 * kk) / CKS2));
#pragma unroll
                for (int i = 0; i < MB2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[k * BM2 + i * 32 + l31], bv, acc2[i], 0, 0, 0);
            }
            if (s + 1 < a.steps2) store2(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- output
    const int gy = ty * TH + wave, gx = tx * TW + l31;
    if (gy >= a.Ho || gx >= a.Wo) return;
    float *__restrict__ ob = a.out + (long)n * a.out_n_stride + (long)gy * a.Wo + gx;
    const long plane = (long)a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < MB2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (co < a.CO) {
                float v = acc2[i][r] + a.b2[co];
                v = v >= 0.0f ? v : 0.1f * v;
                ob[(long)co * plane] = v;
            }
        }
}

template <int K, int S, int MB1, int MB2, int C1>
static bool launch_pair_t(const PairArgs &a_in, hipStream_t s)
{
    using G = PairGeom<K, S, MB1, MB2, C1>;
    static PerDeviceOnce once;
    if (G::lds_bytes > 64 * 1024 &&
        !once.ensure([] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_pair_kernel<K, S, MB1, MB2, C1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes) == hipSuccess; }))
        return false;
    PairArgs a = a_in;
    a.steps1 = (a.Cin + C1 - 1) / C1;
    a.steps2 = (a.CM + G::CKS2 - 1) / G::CKS2;
    dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x));
    hipLaunchKernelGGL((conv_pair_kernel<K, S, MB1, MB2, C1>), grid, dim3(256), G::lds_bytes, s, a);
    return true;
}

// Which pairs the fused kernel serves.  It exists for taps / stride (9,2), (7,2), (3,1) and up to 64 channels on both sides, and is
// USED where it measured faster end to end (batch 32, 256x192): conv1 (6 -> 32 -> 32, 9 taps; 0.156 -> 0.142 ms per pair, +0.6 %
// overall, and its 100 MB intermediate never reaches HBM).  The extra-input pairs (7..9 -> 32 -> 32, 3 taps) are faster fused
// (0.029 -> 0.023 ms) but already hidden on the side stream; with 32+ input channels the first phase dominates and its 9 (or 5)
// pixel blocks do not split evenly over the 4 waves (conv2 0.070 -> 0.084 ms, conv2_1 0.072 -> 0.088 ms).
// DEMON_FUSED_PAIRS_ALL=1 turns every supported pair on (tests, experiments).
bool conv_pair_applies(int k, int stride, int cin, int cm, int co)
{
    static const int all = getenv("DEMON_FUSED_PAIRS_ALL") ? atoi(getenv("DEMON_FUSED_PAIRS_ALL")) : 0;
    const bool shape = (k == 9 && stride == 2) || (k == 7 && stride == 2) || (k == 3 && stride == 1);
    if (!shape || cm > 64 || co > 64 || cm % 8) return false;
    return all || (k == 9 && cin <= 16);
}

void conv_pair_tiles(int Ho, int Wo, int &tiles_y, int &tiles_x)
{
    tiles_y = (Ho + 3) / 4;
    tiles_x = (Wo + 31) / 32;
}

int conv_pair_cks(int k) { return k == 3 ? 8 : 2; }

bool launch_conv_pair(const PairArgs &a, int k, int stride, hipStream_t s)
{
    const int mb1 = (a.CM + 31) / 32, mb2 = (a.CO + 31) / 32;
    // conv1 of every block: 6 input channels, 9 taps -> the whole phase-1 reduction in one K-step
    static const int one_step = getenv("DEMON_PAIR_ONE_STEP") ? atoi(getenv("DEMON_PAIR_ONE_STEP")) : 1;
    if (k == 9 && stride == 2 && a.Cin == 6 && mb1 == 1 && one_step) {
        if (mb2 == 1) return launch_pair_t<9, 2, 1, 1, 6>(a, s);
        return launch_pair_t<9, 2, 1, 2, 6>(a, s);
    }
#define PAIR_CASE(KK, SS, CC)                                                  \
    if (k == KK && stride == SS) {                                             \
        if (mb1 == 1 && mb2 == 1) return launch_pair_t<KK, SS, 1, 1, CC>(a, s); \
        if (mb1 == 1 && mb2 == 2) return launch_pair_t<KK, SS, 1, 2, CC>(a, s); \
        if (mb1 == 2 && mb2 == 1) return launch_pair_t<KK, SS, 2, 1, CC>(a, s); \
        return launch_pair_t<KK, SS, 2, 2, CC>(a, s);                          \
    }
    PAIR_CASE(9, 2, 2)
    PAIR_CASE(7, 2, 2)
    PAIR_CASE(3, 1, 8)
#undef PAIR_CASE
    return false;
}

}  // namespace demon
