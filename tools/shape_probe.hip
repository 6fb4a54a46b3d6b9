// shape_probe.hip - which bf16 MFMA shape leaves more issue room for a service wave on the same SIMD?
// The split kernels pair, on every SIMD, a matrix wave (MFMAs + LDS fragment reads only) with a service wave (splits, products, address arithmetic).
// MI355X_MICROARCH.md: an MFMA holds its SIMD's vector-issue port for 8 cycles - of 16 for v_mfma_f32_16x16x32_bf16, of 32 for v_mfma_f32_32x32x16_bf16 - so at
// equal flops the 32x32 shape should leave the partner wave three quarters of the issue cycles instead of half.  This probe runs the member-gradient kernel's
// matrix-wave work per tile (one product block x 64 columns x 32 hyperedges x K = 128 through six bf16 terms: 192 MFMAs 16x16x32 = 96 MFMAs 32x32x16, weight
// planes resident in 192 registers, fragments read from LDS images) beside a service wave doing SVC fragments' worth of LDS reads + products + splits, one
// barrier per tile, no global traffic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/shape_probe tools/shape_probe.hip && /tmp/shape_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int TE = 32, RB = 256;

__device__ __forceinline__ unsigned pack_hi(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

__device__ __forceinline__ void split8(const float (&x)[8], v4u (&p)[3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const float r0 = x0 - top16(x0), r1 = x1 - top16(x1);
        const float l0 = r0 - top16(r0), l1 = r1 - top16(r1);
        p[0][i] = pack_hi(x0, x1);
        p[1][i] = pack_hi(r0, r1);
        p[2][i] = pack_hi(l0, l1);
    }
}

// SHAPE 16 / 32; WHO: 0 both roles, 1 matrix waves alone, 2 service waves alone; SVC: fragments of service work per thread and tile; PRIO: s_setprio of the service waves
template <int SHAPE, int WHO, int SVC, int PRIO>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ members, const short* __restrict__ wplanes, float* __restrict__ out, int tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char planes[2][3][TE][RB];     // bf16 images of a 32 x 128 tile, chunk c of row r at c ^ (r & 15)
    __shared__ __attribute__((aligned(16))) float tile[2][TE][128];
    __shared__ __attribute__((aligned(16))) float dz[4][TE][68];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 2 * TE * 128; i += 512) (&tile[0][0][0])[i] = members[i];
    for (int i = tid; i < 2 * 3 * TE * RB / 4; i += 512) reinterpret_cast<unsigned*>(&planes[0][0][0][0])[i] = 0x3f803f80u + i;
    __syncthreads();
    if (wave >= 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        const int st = tid - 256, row = st >> 3, o = st & 7;
        v4u keep = v4u{0, 0, 0, 0};
        for (int t = 0; t < tiles; ++t) {
            if (WHO != 1) {
#pragma unroll
                for (int f = 0; f < SVC; ++f) {
                    const int c0 = ((4 * f + (o >> 1)) ^ (row & 15)) & 31;
                    const v4f u0 = *reinterpret_cast<const v4f*>(&tile[0][row][4 * c0]), u1 = *reinterpret_cast<const v4f*>(&tile[0][row][4 * (c0 ^ 1)]);
                    const v4f q0 = *reinterpret_cast<const v4f*>(&tile[1][row][4 * c0]), q1 = *reinterpret_cast<const v4f*>(&tile[1][row][4 * (c0 ^ 1)]);
                    float z[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { z[i] = u0[i] * q0[i] + __uint_as_float(keep[i]); z[4 + i] = u1[i] * q1[i]; }
                    v4u p[3];
                    split8(z, p);
                    keep ^= p[2];
                    if (f < 2) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<v4u*>(&planes[(t + 1) & 1][pl][row][(((o + 8 * f) ^ (row & 15)) & 15) << 4]) = p[pl];
                    }
                }
            }
            __syncthreads();
        }
        if (keep[0] == 0x12345u) out[tid] = 1.f;
        return;
    }
    // matrix waves: weight planes of one block x 64 columns x K = 128 in 192 registers
    v8s w[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) w[i] = *reinterpret_cast<const v8s*>(wplanes + ((wave * 48 + i) * 64 + lane) * 8);
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
    float sink = 0.f;
    for (int t = 0; t < tiles; ++t) {
        if (WHO != 2) {
            const unsigned char* pb = &planes[t & 1][0][0][0];
            if (SHAPE == 16) {
                const int arow = lane & 15, kq = lane >> 4;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    v4f acc[4] = {v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        v8s a[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const v8s*>(pb + p * (TE * RB) + (16 * rt + arow) * RB + (((4 * kb + kq) ^ arow) << 4));
#pragma unroll
                        for (int term = 0; term < 6; ++term)
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct)
                                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(kb * 4 + ct) * 3 + TB[term]], a[TA[term]], acc[ct], 0, 0, 0);
                    }
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<v4f*>(&dz[wave][16 * rt + arow][16 * ct + 4 * kq]) = acc[ct];
                }
            } else {
                // 32x32x16: M = 32 weight columns (2 tiles), N = the tile's 32 hyperedges, K = 16 per step (8 steps); lane: n = lane & 31, k = 8 (lane >> 5) ..
                const int n = lane & 31, kh = lane >> 5;
                v16f acc[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    v8s b[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const v8s*>(pb + p * (TE * RB) + n * RB + (((2 * ks + kh) ^ (n & 15)) << 4));
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(ks * 2 + mt) * 3 + TB[term]], b[TA[term]], acc[mt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<v4f*>(&dz[wave][n][32 * mt + 8 * g + 4 * kh]) = v4f{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
            }
        }
        __syncthreads();
        sink += dz[wave][lane & 31][lane >> 1];
    }
    out[(blockIdx.x * 4 + wave) * 64 + lane] = sink;
}

template <int SHAPE, int WHO, int SVC, int PRIO> float run(const float* members, const short* wplanes, float* out, int tiles) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<SHAPE, WHO, SVC, PRIO><<<256, 512>>>(members, wplanes, out, tiles);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) probe<SHAPE, WHO, SVC, PRIO><<<256, 512>>>(members, wplanes, out, tiles);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

template <int SVC, int PRIO> void table(const float* members, const short* wplanes, float* out, int tiles) {
    const float a16 = run<16, 1, SVC, PRIO>(members, wplanes, out, tiles), a32 = run<32, 1, SVC, PRIO>(members, wplanes, out, tiles);
    const float s = run<16, 2, SVC, PRIO>(members, wplanes, out, tiles);
    const float b16 = run<16, 0, SVC, PRIO>(members, wplanes, out, tiles), b32 = run<32, 0, SVC, PRIO>(members, wplanes, out, tiles);
    printf("{\"service_fragments\": %d, \"service_prio\": %d, \"matrix_alone_16x16x32_ms\": %.4f, \"matrix_alone_32x32x16_ms\": %.4f, \"service_alone_ms\": %.4f, "
           "\"both_16x16x32_ms\": %.4f, \"both_32x32x16_ms\": %.4f}\n", SVC, PRIO, a16, a32, s, b16, b32);
}

int main() {
    const int tiles = 537;                                   // C3: 2 x 68,750 half-tiles over 256 workgroups
    std::vector<float> hm(2 * TE * 128);
    for (auto& v : hm) v = (static_cast<float>(rand()) / RAND_MAX - 0.5f) * 0.4f;
    std::vector<short> hw(4 * 48 * 64 * 8);
    for (auto& v : hw) v = static_cast<short>(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));
    float *members, *out;
    short* wplanes;
    hipMalloc(&members, hm.size() * 4);
    hipMalloc(&wplanes, hw.size() * 2);
    hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(members, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(wplanes, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    table<4, 0>(members, wplanes, out, tiles);
    table<4, 3>(members, wplanes, out, tiles);
    table<8, 0>(members, wplanes, out, tiles);
    table<8, 3>(members, wplanes, out, tiles);
    table<12, 3>(members, wplanes, out, tiles);
    return 0;
}
