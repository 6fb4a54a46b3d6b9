// dot2_probe.hip - does v_dot2c_f32_bf16 give the EXACT remainders of the three-term bf16 split?  r = x - top16(x) as
// dot2({hi_a, hi_b}, {-1, 0}, x_a): one instruction instead of v_and + v_sub.  Compares the planes of both formulations bit for bit over
// random values of many magnitudes (and reports how values near the denormal range behave).  Result on MI355X: identical planes (0 of
// 12.6 M differ) when the selector is a register - as the inline constant -1.0 it acts on both halves - but in the kernels the form was
// SLOWER (forward +2 %, weight gradients +8 %: the dot instruction is not full rate), so split_arith.hip keeps mask-and-subtract.
//   hipcc --offload-arch=gfx950 -O3 -o dot2_probe tools/dot2_probe.hip && ./dot2_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_hi(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

__global__ void both(const float* x, unsigned* ref, unsigned* got, int n_pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    {
        const float ra = a - top16(a), rb = b - top16(b);
        const float la = ra - top16(ra), lb = rb - top16(rb);
        ref[3 * i] = pack_hi(a, b);
        ref[3 * i + 1] = pack_hi(ra, rb);
        ref[3 * i + 2] = pack_hi(la, lb);
    }
    {
        // {-1, 0} must not reach the instruction as the INLINE constant -1.0 (measured: that form subtracts both halves); a register does
        unsigned sa = 0x0000bf80u, sb = 0xbf800000u;
        asm volatile("" : "+s"(sa), "+s"(sb));
        const v2bf sel_a = __builtin_bit_cast(v2bf, sa), sel_b = __builtin_bit_cast(v2bf, sb);
        const unsigned p0 = pack_hi(a, b);
        const float ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, p0), sel_a, a, false);
        const float rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, p0), sel_b, b, false);
        const unsigned p1 = pack_hi(ra, rb);
        const float la = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, p1), sel_a, ra, false);
        const float lb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, p1), sel_b, rb, false);
        got[3 * i] = p0;
        got[3 * i + 1] = p1;
        got[3 * i + 2] = pack_hi(la, lb);
    }
}

int main() {
    const int n_pairs = 1 << 22;
    std::vector<float> hx(2 * n_pairs);
    srand(7);
    for (int i = 0; i < 2 * n_pairs; ++i) {
        const int e = (i % 97 == 0) ? -126 - rand() % 20 : (rand() % 80) - 40;            // every 97th value: at / below the normal range's edge
        const float m = 1.f + static_cast<float>(rand()) / RAND_MAX;
        hx[i] = std::ldexp((rand() & 1) ? m : -m, e);
    }
    float* x;
    unsigned *ref, *got;
    (void)hipMalloc(&x, hx.size() * 4);
    (void)hipMalloc(&ref, 3ull * n_pairs * 4);
    (void)hipMalloc(&got, 3ull * n_pairs * 4);
    (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    both<<<n_pairs / 256, 256>>>(x, ref, got, n_pairs);
    std::vector<unsigned> hr(3ull * n_pairs), hg(3ull * n_pairs);
    (void)hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hg.data(), got, hg.size() * 4, hipMemcpyDeviceToHost);
    long bad_normal = 0, bad_tiny = 0;
    for (int i = 0; i < n_pairs; ++i) {
        const bool tiny = std::fabs(hx[2 * i]) < 1e-30f || std::fabs(hx[2 * i + 1]) < 1e-30f;
        for (int p = 0; p < 3; ++p)
            if (hr[3 * i + p] != hg[3 * i + p]) {
                (tiny ? bad_tiny : bad_normal)++;
                if ((tiny ? bad_tiny : bad_normal) <= 3)
                    printf("differs: x = (%g, %g) plane %d and-sub %08x dot2 %08x\n", hx[2 * i], hx[2 * i + 1], p, hr[3 * i + p], hg[3 * i + p]);
            }
    }
    printf("{\"pairs\": %d, \"planes_differing_normal_range\": %ld, \"planes_differing_below_1e-30\": %ld}\n", n_pairs, bad_normal, bad_tiny);
    return bad_normal != 0;
}
