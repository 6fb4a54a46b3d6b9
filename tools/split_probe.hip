// split_probe.hip - micro-benchmark behind DESIGN.md's "fp32 through three bf16 terms" section: what one CU sustains when the
// interact forward's contraction runs as bf16 MFMAs over exactly split fp32 operands (x = hi + mid + lo, three bf16 values whose
// sum IS x; six of the nine partial products carry everything above 2^-26 |a||b|).  No global traffic inside the loop: the member
// tile sits in LDS, the weight planes in registers, so the figure is the matrix-pipe + vector-issue bound of that formulation.
//   hipcc --offload-arch=gfx950 -O3 -o split_probe tools/split_probe.hip && ./split_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int TE = 32, D = 128;

__device__ __forceinline__ unsigned pack_hi(float a, float b) {          // {b.hi16, a.hi16}
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

struct Planes { v8s p[3]; };

__device__ __forceinline__ Planes split8(const float (&x)[8]) {
    v4u a, b, c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const float r0 = x0 - top16(x0), r1 = x1 - top16(x1);
        const float l0 = r0 - top16(r0), l1 = r1 - top16(r1);
        a[i] = pack_hi(x0, x1);
        b[i] = pack_hi(r0, r1);
        c[i] = pack_hi(l0, l1);
    }
    Planes out;
    out.p[0] = __builtin_bit_cast(v8s, a);
    out.p[1] = __builtin_bit_cast(v8s, b);
    out.p[2] = __builtin_bit_cast(v8s, c);
    return out;
}

// MODE 1: MFMAs only.  MODE 3: LDS reads + products + split + MFMAs.  MODE 4: MODE 3 + one barrier and a partial-sum image per tile.
template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ members, const short* __restrict__ wplanes, float* __restrict__ out, int tiles) {
    __shared__ __attribute__((aligned(16))) float tile[3][TE][D];
    __shared__ __attribute__((aligned(16))) float part[8][16][68];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int blk = wave >> 1, ch = wave & 1;
    for (int i = tid; i < 3 * TE * D; i += 512) (&tile[0][0][0])[i] = members[i];
    v8s w[2][4][3];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                w[kb][ct][pl] = *reinterpret_cast<const v8s*>(wplanes + ((((wave * 2 + kb) * 4 + ct) * 3 + pl) * 64 + lane) * 8);
    __syncthreads();
    v4f acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = v4f{0.f, 0.f, 0.f, 0.f};
    const int arow = lane & 15, kq = lane >> 4;
    Planes fixed;
    {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = tile[0][arow][8 * kq + i];
        fixed = split8(x);
    }
    const int m0 = blk == 0 ? 0 : (blk == 1 ? 1 : (blk == 2 ? 2 : 0));
    const int m1 = blk == 0 ? 1 : (blk == 1 ? 2 : (blk == 2 ? 0 : 1));
    if (MODE == 5 || MODE == 6 || MODE == 7) {
        // wave-specialised: waves 0-3 issue MFMAs only (192 per tile), waves 4-7 only products + splits (8 fragments' worth per tile);
        // MODE 6: the MFMA waves alone, MODE 7: the split waves alone
        float sink = 0.f;
        v8s keep = fixed.p[0];
        for (int t = 0; t < tiles; ++t) {
            asm volatile("" ::: "memory");
            if (wave < 4) {
                if (MODE != 7) {
#pragma unroll
                    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int kb = 0; kb < 2; ++kb) {
                                constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                                for (int term = 0; term < 6; ++term)
#pragma unroll
                                    for (int ct = 0; ct < 4; ++ct)
                                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fixed.p[TA[term]], w[kb][ct][TB[term]], acc[rt][ct], 0, 0, 0);
                            }
                }
            } else if (MODE != 6) {
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    const int row = 16 * (f & 1) + arow;
                    const int c4 = (8 * f + 8 * kq) >> 2 & 31;
                    const int s0 = ((c4 ^ (row & 15)) << 2) & 127, s1 = (((c4 + 1) ^ (row & 15)) << 2) & 127;
                    const v4f u0 = *reinterpret_cast<const v4f*>(&tile[0][row][s0]), u1 = *reinterpret_cast<const v4f*>(&tile[0][row][s1]);
                    const v4f q0 = *reinterpret_cast<const v4f*>(&tile[1][row][s0]), q1 = *reinterpret_cast<const v4f*>(&tile[1][row][s1]);
                    float z[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { z[i] = u0[i] * q0[i]; z[4 + i] = u1[i] * q1[i]; }
                    const Planes a = split8(z);
                    keep ^= a.p[0] ^ a.p[1] ^ a.p[2];
                }
                *reinterpret_cast<v8s*>(&part[wave][arow][4 * kq]) = keep;
            }
        }
        if (sink == 1.f) out[1] = sink;
    } else
    for (int t = 0; t < tiles; ++t) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Planes a = fixed;
                if (MODE >= 3) {
                    const int row = 16 * rt + arow;
                    const int c4 = (64 * ch + 32 * kb + 8 * kq) >> 2;              // 16-byte chunk, two consecutive ones
                    const int s0 = ((c4 ^ (row & 15)) << 2), s1 = (((c4 + 1) ^ (row & 15)) << 2);
                    const v4f u0 = *reinterpret_cast<const v4f*>(&tile[m0][row][s0]), u1 = *reinterpret_cast<const v4f*>(&tile[m0][row][s1]);
                    const v4f q0 = *reinterpret_cast<const v4f*>(&tile[m1][row][s0]), q1 = *reinterpret_cast<const v4f*>(&tile[m1][row][s1]);
                    float z[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { z[i] = u0[i] * q0[i]; z[4 + i] = u1[i] * q1[i]; }
                    if (blk == 3) {
                        const v4f i0 = *reinterpret_cast<const v4f*>(&tile[2][row][s0]), i1 = *reinterpret_cast<const v4f*>(&tile[2][row][s1]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { z[i] *= i0[i]; z[4 + i] *= i1[i]; }
                    }
                    a = split8(z);
                }
                constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[TA[term]], w[kb][ct][TB[term]], acc[rt][ct], 0, 0, 0);
            }
        if (MODE >= 4) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][4 * kq + r][16 * ct + arow] = acc[0][ct][r] + acc[1][ct][r];
            __syncthreads();
            float s = 0.f;
            const int row = tid >> 5, col = (tid & 31) * 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += part[k][row][col] + part[k][row][col + 1];
            if (s == 12345.678f) out[0] = s;
        }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<v4f*>(out + (((blockIdx.x * 8 + wave) * 8 + rt * 4 + ct) * 64 + lane) * 4) = acc[rt][ct];
}

template <int MODE> float run(const float* members, const short* wplanes, float* out, int tiles) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<MODE><<<256, 512>>>(members, wplanes, out, tiles);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) probe<MODE><<<256, 512>>>(members, wplanes, out, tiles);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const int tiles = 537;                                   // C3: 2 x 68,750 half-tiles over 256 workgroups
    std::vector<float> hm(3 * TE * D);
    for (auto& v : hm) v = (static_cast<float>(rand()) / RAND_MAX - 0.5f) * 0.4f;
    std::vector<short> hw(8 * 2 * 4 * 3 * 64 * 8);
    for (auto& v : hw) v = static_cast<short>(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));
    float *members, *out;
    short* wplanes;
    hipMalloc(&members, hm.size() * 4);
    hipMalloc(&wplanes, hw.size() * 2);
    hipMalloc(&out, 256 * 8 * 8 * 64 * 4 * 4);
    hipMemcpy(members, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(wplanes, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    const float m1 = run<1>(members, wplanes, out, tiles), m3 = run<3>(members, wplanes, out, tiles), m4 = run<4>(members, wplanes, out, tiles);
    const float m5 = run<5>(members, wplanes, out, tiles), m6 = run<6>(members, wplanes, out, tiles), m7 = run<7>(members, wplanes, out, tiles);
    printf("{\"specialised_both_ms\": %.4f, \"mfma_waves_alone_ms\": %.4f, \"split_waves_alone_ms\": %.4f}\n", m5, m6, m7);
    const double mfma = 256.0 * 8 * tiles * 96;             // MFMAs per launch
    printf("{\"tiles_per_workgroup\": %d, \"mfma_only_ms\": %.4f, \"with_split_ms\": %.4f, \"with_split_and_barrier_ms\": %.4f, "
           "\"mfma_only_cycles_per_mfma_per_simd_at_2.1GHz\": %.2f}\n",
           tiles, m1, m3, m4, m1 * 1e-3 * 2.1e9 / (mfma / 1024));
    return 0;
}
