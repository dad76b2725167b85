// Inference stem of SimpleBaseline-ResNet as ONE launch: conv1 7x7 / stride 2 / pad 3 (3 -> 64 channels) + the eval-mode
// BatchNorm affine + ReLU + MaxPool2d(3, 2, 1)   (src/modeling/simplebaseline/pose_resnet.py:151-156 and its forward:
// x = maxpool(relu(bn1(conv1(x))))).  The 64-channel convolution output -- the largest activation of the network, 1.2 GB at
// BASELINE.json configs[4] -- is never written: the tiled launches it replaces moved 2.4 GB + the 9-fold tap gather for it.
//
// Direct form (no implicit-GEMM gather): the input lives as zero-padded NHWC4 (engine.Plan._c_stem), so the K run of one
// kernel ROW is 8 pixels x 4 channels = 64 contiguous bytes.  A workgroup is persistent and keeps the WHOLE weight tensor
// in registers as MFMA "A" fragments (4 channel tiles x 7 kernel rows); per tile of 8 x 8 pooled pixels it
//   1. stages the 39 x 40-pixel input patch of the 17 x 17 convolution outputs the tile's windows touch in LDS (12 KB; the
//      patch of the NEXT tile is requested before this one is multiplied),
//   2. forms every B fragment straight from the patch (one ds_read_b128 per 16 outputs and kernel row: the seven rows of a
//      window are seven row offsets into the patch), 28 MFMAs (v_mfma_f32_16x16x32) per 16 outputs,
//   3. applies the per-channel affine + ReLU to the fp32 accumulators, rounds to the storage type and parks the 17 x 17 x 64
//      tile in LDS (outputs outside the image as 0: after the ReLU the maximum over a window's valid positions is unchanged),
//   4. takes the 3 x 3 / stride 2 maxima from LDS and writes full 128-byte NHWC rows of the pooled tensor.
// K order (kernel row ascending, 32 elements per MFMA) and epilogue arithmetic are those of igemm_ring_kernel on the same
// pack, and the maximum is taken over the ROUNDED values: the result is bit-identical to conv -> maxpool as two launches.
#include "common.h"
#include "multi.h"

template <typename T> struct StemMma;
template <> struct StemMma<bf16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct StemMma<f16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

struct StemPoolArgs {
    const unsigned char* img;    // [n][hp][wp][4] T: image pixel (y, x) at (y + 3, x + 3), borders zero
    const unsigned char* w;      // weight pack [64 (padded to 128)][7 kernel rows][64 elements] T (engine.Plan._c_stem)
    const float* bias;
    const float* scale;          // eval-mode BatchNorm folded: out = acc * scale + (bias * scale + shift)
    const float* shift;
    unsigned char* out;          // [n][ph][pw][64] T
    int n, hp, wp;               // padded image
    int ch, cw;                  // convolution output size
    int ph, pw;                  // pooled output size
    int ty, tx, ntiles;          // tiles of 8 x 8 pooled pixels per image (rows, columns), in all
};

constexpr int SP_CT = 17;                    // convolution outputs per tile side
constexpr int SP_NPX = SP_CT * SP_CT;        // 289
constexpr int SP_GROUPS = (SP_NPX + 15) / 16;
constexpr int SP_PROWS = 2 * (SP_CT - 1) + 7;            // 39 input rows
constexpr int SP_PPITCH = (2 * (SP_CT - 1) + 8) * 8;     // 40 pixels x 8 bytes = 320 bytes per patch row
constexpr int SP_PCHUNKS = SP_PROWS * (SP_PPITCH / 16);  // 780 16-byte chunks
constexpr int SP_CPITCH = 64 * 2 + 8;                    // convolution-tile row pitch in LDS (bank shift per pixel)
constexpr int SP_PATCH_BYTES = SP_PROWS * SP_PPITCH;     // 12 480
constexpr int SP_CST_OFF = (SP_PATCH_BYTES + SP_NPX * SP_CPITCH + 15) / 16 * 16;
constexpr int SP_LDS = SP_CST_OFF + 2 * 64 * 4;                      // + scale / shift of the 64 channels

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemPoolArgs p) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* ctile = smem + SP_PATCH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, pl = lane & 15;

    // weights -> registers: A fragment of channel tile i, kernel row r = chunk q of pack row (16 i + pl, r)
    uint4 W[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r)
            W[i][r] = *reinterpret_cast<const uint4*>(p.w + ((long)((16 * i + pl) * 7 + r) * 64 + q * 8) * 2);
    // per-channel affine (scale, shift incl. the bias) in LDS: the epilogue of a group reads the lane's 16 channels from there
    // (kept in registers they pushed the kernel into scratch)
    float* cst = reinterpret_cast<float*>(smem + SP_CST_OFF);
    if (tid < 64) {
        const float sc = p.scale ? p.scale[tid] : 1.f;
        cst[tid] = sc;
        cst[64 + tid] = (p.bias ? p.bias[tid] : 0.f) * sc + (p.scale ? p.shift[tid] : 0.f);
    }
    const int tiles_per_img = p.ty * p.tx;
    const long img_bytes = (long)p.hp * p.wp * 8;
    const int row_bytes = p.wp * 8;
    // the lane's patch chunks: chunk k = tid + 256 j -> (patch row, 16-byte column)
    constexpr int NCH = (SP_PCHUNKS + 255) / 256;
    auto load_patch = [&](int tile, uint4 (&R)[NCH]) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int tyi = t2 / p.tx, txi = t2 - tyi * p.tx;
        const int iy0 = 32 * tyi - 2, ixb0 = (32 * txi - 2) * 8;           // first input row / first byte of the patch in the padded image
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int k = tid + 256 * j;
            const int pr = k / (SP_PPITCH / 16), pc = k - pr * (SP_PPITCH / 16);
            const int iy = iy0 + pr, ib = ixb0 + pc * 16;
            const bool ok = tile < p.ntiles && k < SP_PCHUNKS && (unsigned)iy < (unsigned)p.hp && ib >= 0 && ib + 16 <= row_bytes;
            R[j] = ok ? *reinterpret_cast<const uint4*>(p.img + (long)b * img_bytes + (long)iy * row_bytes + ib) : uint4{0u, 0u, 0u, 0u};
        }
    };
    uint4 R[NCH];
    int tile = blockIdx.x;
    load_patch(tile, R);
    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int tyi = t2 / p.tx, txi = t2 - tyi * p.tx;
        const int cy0 = 16 * tyi - 1, cx0 = 16 * txi - 1;                     // convolution output of local (0, 0)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int k = tid + 256 * j;
            if (k < SP_PCHUNKS) *reinterpret_cast<uint4*>(patch + k * 16) = R[j];
        }
        __syncthreads();
        load_patch(tile + gridDim.x, R);                                      // flies under the MFMAs of this tile

        for (int g = wave; g < SP_GROUPS; g += 4) {
            const int idx = g * 16 + pl;
            const int idc = idx < SP_NPX ? idx : SP_NPX - 1;
            const int ly = idc / SP_CT, lx = idc - ly * SP_CT;
            const unsigned char* bsrc = patch + (2 * ly) * SP_PPITCH + lx * 16 + q * 16;
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            uint4 B[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) B[r] = *reinterpret_cast<const uint4*>(bsrc + r * SP_PPITCH);
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) StemMma<T>::run(W[i][r], B[r], acc[i]);
            const bool inside = idx < SP_NPX && (unsigned)(cy0 + ly) < (unsigned)p.ch && (unsigned)(cx0 + lx) < (unsigned)p.cw;
            if (idx < SP_NPX) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    union { uint2 u; T e[4]; } pk;
                    const float4 s4 = *reinterpret_cast<const float4*>(cst + 16 * i + 4 * q);
                    const float4 b4 = *reinterpret_cast<const float4*>(cst + 64 + 16 * i + 4 * q);
                    const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = fmaxf(acc[i][j] * sv[j] + bv[j], 0.f);
                        pk.e[j] = from_f<T>(inside ? v : 0.f);
                    }
                    pk.u.x &= 0x7fff7fffu; pk.u.y &= 0x7fff7fffu;             // -0 -> +0 (see the maximum below)
                    *reinterpret_cast<uint2*>(ctile + idx * SP_CPITCH + (16 * i + 4 * q) * 2) = pk.u;
                }
            }
        }
        __syncthreads();
        // 64 pooled pixels x 8 chunks of 8 channels: two items per thread
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 256 * k;
            const int chunk = item & 7, pp = item >> 3;
            const int ppy = pp >> 3, ppx = pp & 7;
            const int oy = 8 * tyi + ppy, ox = 8 * txi + ppx;
            // every stored value is >= +0 (ReLU; the sign bit is cleared at the store) or a NaN: the order of such 16-bit floats
            // is the order of their bit patterns as unsigned integers, NaN patterns above every number -- the window maximum,
            // with the framework's "a NaN wins" rule, is a packed unsigned 16-bit maximum (4 instructions per tap instead of ~40)
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 best = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ly = 2 * ppy + t / 3, lx = 2 * ppx + t % 3;
                const unsigned char* src = ctile + (ly * SP_CT + lx) * SP_CPITCH + chunk * 16;
                const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
                best = __builtin_elementwise_max(best, __builtin_bit_cast(u16x8, uint4{lo.x, lo.y, hi.x, hi.y}));
            }
            if (oy < p.ph && ox < p.pw)
                *reinterpret_cast<uint4*>(p.out + (((long)b * p.ph + oy) * p.pw + ox) * 128 + chunk * 16) = __builtin_bit_cast(uint4, best);
        }
        // the next iteration's patch writes touch only `patch` (its reads ended at the barrier above); its tile writes come
        // after the next barrier, i.e. after every thread has finished these reads
    }
}

extern "C" int lh_stem_pool(const void* img, int n, int hp, int wp, const void* wpack, const float* bias, const float* scale,
                            const float* shift, void* out, int conv_h, int conv_w, int relu, int dtype, void* stream) {
    LH_REQUIRE(img && wpack && out && n > 0 && conv_h > 0 && conv_w > 0, "lh_stem_pool: bad arguments");
    LH_REQUIRE(dtype == LH_BF16 || dtype == LH_F16, "lh_stem_pool: 16-bit types only (dtype %d)", dtype);
    LH_REQUIRE(relu, "lh_stem_pool: the fused maximum treats positions outside the image as 0, which needs the ReLU in front of the pool");
    LH_REQUIRE((scale == nullptr) == (shift == nullptr), "lh_stem_pool: scale and shift must come together");
    LH_REQUIRE(hp >= 2 * (conv_h - 1) + 7 && wp >= 2 * (conv_w - 1) + 8, "lh_stem_pool: padded image %d x %d too small for %d x %d outputs", hp, wp, conv_h, conv_w);
    StemPoolArgs a;
    a.img = (const unsigned char*)img; a.w = (const unsigned char*)wpack; a.bias = bias; a.scale = scale; a.shift = shift;
    a.out = (unsigned char*)out; a.n = n; a.hp = hp; a.wp = wp; a.ch = conv_h; a.cw = conv_w;
    a.ph = (conv_h + 2 - 3) / 2 + 1; a.pw = (conv_w + 2 - 3) / 2 + 1;
    a.ty = ceil_div(a.ph, 8); a.tx = ceil_div(a.pw, 8);
    const long nt = (long)n * a.ty * a.tx;
    LH_REQUIRE(nt < (1L << 30), "lh_stem_pool: too many tiles");
    a.ntiles = (int)nt;
    const int grid = (int)(nt < 512 ? nt : 512);                      // persistent: two workgroups per CU
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LH_BF16) hipLaunchKernelGGL((stem_pool_kernel<bf16>), dim3(grid), dim3(256), SP_LDS, s, a);
    else hipLaunchKernelGGL((stem_pool_kernel<f16>), dim3(grid), dim3(256), SP_LDS, s, a);
    LH_LAUNCH_CHECK("stem_pool launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ training stem
// conv1 alone, for training plans (pose_resnet.py:151-152: conv1 output feeds bn1 in training mode): the same direct form
// -- weights in registers, the input patch of a tile staged in LDS, B fragments straight from it -- on tiles of 16 x 16
// convolution outputs, writing the RAW convolution output in full 128-byte NHWC rows and the per-channel sums / sums of
// squares of the STORED (rounded) values, one statistics row per workgroup ([grid][2][64], the layout lh_bn_finalize
// reads).  The tiled kernel it replaces ran 8 192 workgroups of seven ring stages each for this layer: prologue and
// epilogue latency, not work (96 us for 134 MB of output at batch 64).  K order = kernel row ascending, 32 elements per
// MFMA: the convolution output is bit-identical to the tiled kernel's; the statistics differ in the order of their fp32
// partial sums only.
struct StemConvArgs {
    const unsigned char* img;
    const unsigned char* w;
    unsigned char* out;          // [n][ch][cw][64] T
    float* stats;                // [gridDim.x][2][64]
    int n, hp, wp, ch, cw, ty, tx, ntiles;
};
constexpr int SC_CT = 16, SC_NPX = 256, SC_GROUPS = 16;
constexpr int SC_PROWS = 2 * (SC_CT - 1) + 7;            // 37 input rows
constexpr int SC_PPITCH = (2 * (SC_CT - 1) + 8) * 8;     // 38 pixels x 8 bytes = 304 bytes per patch row
constexpr int SC_PCHUNKS = SC_PROWS * (SC_PPITCH / 16);  // 703 16-byte chunks
constexpr int SC_PATCH_BYTES = (SC_PROWS * SC_PPITCH + 15) / 16 * 16;
constexpr int SC_CPITCH = 64 * 2 + 8;
constexpr int SC_LDS = SC_PATCH_BYTES + SC_NPX * SC_CPITCH;          // 11 248 + 34 816

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const StemConvArgs p) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* ctile = smem + SC_PATCH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, pl = lane & 15;
    uint4 W[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r)
            W[i][r] = *reinterpret_cast<const uint4*>(p.w + ((long)((16 * i + pl) * 7 + r) * 64 + q * 8) * 2);
    const int tiles_per_img = p.ty * p.tx;
    const long img_bytes = (long)p.hp * p.wp * 8;
    const int row_bytes = p.wp * 8;
    constexpr int NCH = (SC_PCHUNKS + 255) / 256;
    auto load_patch = [&](int tile, uint4 (&R)[NCH]) {
        const int tl = tile < p.ntiles ? tile : 0;                           // every lane loads (a tile of the problem): the value is not used
        const int b = tl / tiles_per_img, t2 = tl - b * tiles_per_img;
        const int tyi = t2 / p.tx, txi = t2 - tyi * p.tx;
        const int iy0 = 32 * tyi, ixb0 = 32 * txi * 8;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int k = tid + 256 * j;
            const int pr = k / (SC_PPITCH / 16), pc = k - pr * (SC_PPITCH / 16);
            const int iy = iy0 + pr, ib = ixb0 + pc * 16;
            const bool ok = k < SC_PCHUNKS && iy < p.hp && ib + 16 <= row_bytes;
            const uint4 v = *reinterpret_cast<const uint4*>(p.img + (long)b * img_bytes + (long)(ok ? iy : 0) * row_bytes + (ok ? ib : 0));
            R[j] = ok ? v : uint4{0u, 0u, 0u, 0u};
        }
    };
    // statistics of this thread's 8 channels (chunk tid & 7 of every pixel it stores), over all tiles of the workgroup
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    uint4 R[NCH];
    int tile = blockIdx.x;
    load_patch(tile, R);
    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int tyi = t2 / p.tx, txi = t2 - tyi * p.tx;
        const int cy0 = 16 * tyi, cx0 = 16 * txi;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int k = tid + 256 * j;
            if (k < SC_PCHUNKS) *reinterpret_cast<uint4*>(patch + k * 16) = R[j];
        }
        __syncthreads();
        load_patch(tile + gridDim.x, R);                                      // flies under the MFMAs of this tile
        for (int g = wave; g < SC_GROUPS; g += 4) {
            const int idx = g * 16 + pl;
            const int ly = idx >> 4, lx = idx & 15;
            const unsigned char* bsrc = patch + (2 * ly) * SC_PPITCH + lx * 16 + q * 16;
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            uint4 B[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) B[r] = *reinterpret_cast<const uint4*>(bsrc + r * SC_PPITCH);
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) StemMma<T>::run(W[i][r], B[r], acc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                union { uint2 u; T e[4]; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.e[j] = from_f<T>(acc[i][j] * 1.f + 0.f);      // the tiled kernel's epilogue arithmetic without bias / affine
                *reinterpret_cast<uint2*>(ctile + idx * SC_CPITCH + (16 * i + 4 * q) * 2) = pk.u;
            }
        }
        __syncthreads();
        // 256 pixels x 8 chunks: eight items per thread, chunk = tid & 7; full 128-byte lines per 8 consecutive threads
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = tid + 256 * k;
            const int chunk = item & 7, pp = item >> 3;
            const int ly = pp >> 4, lx = pp & 15;
            const int oy = cy0 + ly, ox = cx0 + lx;
            const unsigned char* src = ctile + pp * SC_CPITCH + chunk * 16;
            const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
            const uint4 u = uint4{lo.x, lo.y, hi.x, hi.y};
            if (oy < p.ch && ox < p.cw) {
                float v[8];
                unpack16<T>(u, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
                *reinterpret_cast<uint4*>(p.out + (((long)b * p.ch + oy) * p.cw + ox) * 128 + chunk * 16) = u;
            }
        }
        // the next iteration's patch writes touch only `patch`; its tile writes come after its first barrier, i.e. after every
        // thread has finished these reads
    }
    // one statistics row per workgroup: the 32 threads that share a chunk fold through LDS in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(ctile);                             // [32 row groups][2][64]
    const int chunk = tid & 7, grp = tid >> 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[(grp * 2 + 0) * 64 + chunk * 8 + e] = s1[e];
        red[(grp * 2 + 1) * 64 + chunk * 8 + e] = s2[e];
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        float a = 0.f;
#pragma unroll 4
        for (int r = 0; r < 32; ++r) a += red[(r * 2 + which) * 64 + c];
        p.stats[((long)blockIdx.x * 2 + which) * 64 + c] = a;
    }
}

// rows of the statistics slab lh_stem_conv writes (= its grid): persistent, two workgroups per CU at most
extern "C" int lh_stem_conv_rows(int n, int conv_h, int conv_w) {
    const long nt = (long)n * ceil_div(conv_h, 16) * ceil_div(conv_w, 16);
    return (int)(nt < 512 ? nt : 512);
}

extern "C" int lh_stem_conv(const void* img, int n, int hp, int wp, const void* wpack, void* out, float* stats, int conv_h, int conv_w,
                            int dtype, void* stream) {
    LH_REQUIRE(img && wpack && out && stats && n > 0 && conv_h > 0 && conv_w > 0, "lh_stem_conv: bad arguments");
    LH_REQUIRE(dtype == LH_BF16 || dtype == LH_F16, "lh_stem_conv: 16-bit types only (dtype %d)", dtype);
    LH_REQUIRE(hp >= 2 * (conv_h - 1) + 7 && wp >= 2 * (conv_w - 1) + 8, "lh_stem_conv: padded image %d x %d too small for %d x %d outputs", hp, wp, conv_h, conv_w);
    StemConvArgs a;
    a.img = (const unsigned char*)img; a.w = (const unsigned char*)wpack; a.out = (unsigned char*)out; a.stats = stats;
    a.n = n; a.hp = hp; a.wp = wp; a.ch = conv_h; a.cw = conv_w;
    a.ty = ceil_div(conv_h, 16); a.tx = ceil_div(conv_w, 16);
    const long nt = (long)n * a.ty * a.tx;
    LH_REQUIRE(nt < (1L << 30), "lh_stem_conv: too many tiles");
    a.ntiles = (int)nt;
    const int grid = lh_stem_conv_rows(n, conv_h, conv_w);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LH_BF16) hipLaunchKernelGGL((stem_conv_kernel<bf16>), dim3(grid), dim3(256), SC_LDS, s, a);
    else hipLaunchKernelGGL((stem_conv_kernel<f16>), dim3(grid), dim3(256), SC_LDS, s, a);
    LH_LAUNCH_CHECK("stem_conv launch");
    return LH_OK;
}
