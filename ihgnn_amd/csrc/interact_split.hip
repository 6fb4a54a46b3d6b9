// interact_split.hip - the interactive step's contractions (d = 128, order 3) on the bf16 matrix pipe at fp32 accuracy.
//
// Every fp32 operand x is taken apart EXACTLY into three bf16 terms, x = hi + mid + lo (hi = the top 16 bits of x, mid = the top
// 16 bits of x - hi, lo = the rest: 8 + 8 + 8 significand bits, both subtractions exact), and a product a b is accumulated in fp32
// as the six partial products hi hi + hi mid + mid hi + hi lo + mid mid + lo hi, smallest first.  Each partial product of two
// bf16 values is exact in fp32; the three that are left out are below 2^-26 |a b|, a quarter of the rounding error of ONE fp32
// multiply-add.  Measured against fp64 the result is as close as the fp32-MFMA kernels' (tests/test_gpu_parity.py holds both to
// the same 1e-5 bar, tools/split_probe.hip has the standalone rate measurement): v_mfma_f32_16x16x32_bf16 runs 16 x the rate of
// v_mfma_f32_16x16x4_f32, six of them replace eight -> the same contraction in ~ 0.4 of the matrix-pipe time.
//
// Three bf16 planes of the weights are 1.5 x their fp32 size: 384 KB, more than one workgroup's registers can keep beside the
// accumulators.  So a workgroup is FOUR waves and owns a QUARTER of the output columns (96 weight registers per wave), two
// workgroups share a CU and overlap each other's load / matrix / store phases; the four quarters of one tile range sit on one
// XCD (workgroups are dealt to XCDs round-robin by their linear id), so three of the four reads of a streamed row hit that L2.
#include <cstdlib>

#include "common.hpp"
#include "split.hpp"

namespace {

typedef short v8s __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int kSplitTE = 32;            // hyperedges per tile
constexpr int kSplitRanges = 128;       // contiguous tile ranges (x 4 column quarters = 512 workgroups, two per CU)
constexpr int kSplitThreads = 256;

__device__ __forceinline__ unsigned pack_hi(float a, float b) {          // {top half of b, top half of a}
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// eight consecutive k of one row / column -> the three bf16 planes of that MFMA fragment
struct Planes {
    v4u p[3];
};
__device__ __forceinline__ Planes split8(v4f x0, v4f x1) {
    Planes out;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const v4f x = half == 0 ? x0 : x1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float a = x[2 * i], b = x[2 * i + 1];
            const float ra = a - top16(a), rb = b - top16(b);
            const float la = ra - top16(ra), lb = rb - top16(rb);
            out.p[0][2 * half + i] = pack_hi(a, b);
            out.p[1][2 * half + i] = pack_hi(ra, rb);
            out.p[2][2 * half + i] = pack_hi(la, lb);
        }
    }
    return out;
}

// partial products in accumulation order (A plane, B plane): smallest first
__device__ constexpr int kTermA[6] = {0, 2, 1, 0, 1, 0};
__device__ constexpr int kTermB[6] = {2, 0, 1, 1, 0, 0};

// planes of the member-gradient contraction dz_b[e][c] = sum_j dout[e][j] W[j][(3+b)d + c]   (k runs along j):
// wsp[qtr][b][kb][ct][plane][lane][8], element i = plane of W[32 kb + 8 (lane>>4) + i][(3+b) d + 32 qtr + 16 ct + (lane&15)]
__global__ __launch_bounds__(kBlockThreads) void pack_planes_members_kernel(const float* __restrict__ w, int64_t ld_w, v4u* __restrict__ wsp) {
    constexpr int D = 128, NBLK = 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * NBLK * 4 * 2 * kWave) return;
    const int lane = idx & 63, ct = (idx >> 6) & 1, kb = (idx >> 7) & 3, b = (idx >> 9) & 3, qtr = idx >> 11;
    const float* src = w + static_cast<int64_t>(32 * kb + 8 * (lane >> 4)) * ld_w + (3 + b) * D + 32 * qtr + 16 * ct + (lane & 15);
    v4f x0, x1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x0[i] = src[i * ld_w];
        x1[i] = src[(4 + i) * ld_w];
    }
    const Planes pl = split8(x0, x1);
#pragma unroll
    for (int p = 0; p < 3; ++p) wsp[(static_cast<int64_t>(idx >> 6) * 3 + p) * kWave + lane] = pl.p[p];
}

// Member gradients.  Per tile of 32 hyperedges: the dout rows come in through registers (each thread 16 values, requested two tiles
// ahead), are split and laid down as three bf16 images (16-byte chunk o of row r at o ^ (r & 15): conflict-free ds_read_b128 for the A
// fragments); wave b contracts them with block b's weights for the quarter's 32 columns (96 MFMAs), the four blocks meet in an LDS
// image, and every thread applies the product rule to 4 columns of one hyperedge with the member values it requested before the
// matrix phase.  The split of tile k + 1 is interleaved, one vector instruction per MFMA, with the matrix phase of tile k (the
// images are double-buffered): an MFMA leaves half of its 16 issue cycles to the vector unit, work placed there is almost free,
// work placed between matrix phases is not.  Two barriers per tile; the CU's other workgroup fills the matrix pipe meanwhile.
// UR (hyperedges numbered by user): the user-slot gradient is not stored per hyperedge - wave 3 (one lane per column) scans the
// tile's rows in order inside the next tile's matrix phase (straight-line code: a run start only resets the running sum through a
// scalar factor) and then stores the few finished runs (interact.hip describes the scheme and its boundary table, which is
// indexed by tile range here and shared by the four column quarters).
template <bool UR>
__global__ __launch_bounds__(kSplitThreads, 2) void interact_bwd_members_split_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const v4u* __restrict__ wsp, const float* __restrict__ dout,
    int64_t ld_dout, float* __restrict__ g_out, int64_t n_edges, float* __restrict__ dh_user, int64_t ld_dh, float* __restrict__ bnd_val,
    int32_t* __restrict__ bnd_user) {
    constexpr int TE = kSplitTE, D = 128, QC = 32, DZ = QC + 4;
    __shared__ __attribute__((aligned(16))) unsigned char planes[2][3][TE][256];
    __shared__ __attribute__((aligned(16))) float dzimg[4][TE][DZ];
    __shared__ __attribute__((aligned(16))) float utile[UR ? TE : 1][DZ];
    __shared__ int ids[4][3 * TE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int qtr = (bid >> 3) & 3, range = (bid & 7) + 8 * (bid >> 5);
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t per = (n_tiles + kSplitRanges - 1) / kSplitRanges;
    const int64_t t0 = range * per;
    const int n_my = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(per, n_tiles - t0)));
    if (n_my == 0) {
        if (UR && qtr == 0 && tid == 0) bnd_user[2 * range] = bnd_user[2 * range + 1] = -1;
        return;
    }

    v8s wreg[4][2][3];
    {
        const v4u* wf = wsp + static_cast<int64_t>((qtr * 4 + wave) * 24) * kWave + lane;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int p = 0; p < 3; ++p) wreg[kb][ct][p] = __builtin_bit_cast(v8s, wf[((kb * 2 + ct) * 3 + p) * kWave]);
    }

    const int row = tid >> 3, o = tid & 7;                                // staging / epilogue role: hyperedge row, 16-byte chunk
    auto fetch_id = [&](int k) {
        const int64_t pos = (t0 + k) * (3 * TE) + tid;
        return pos < n_edges * 3 ? i3[pos] : 0;
    };
    auto load_dout = [&](int k, v4f (&dr)[4]) {
#ifdef IHG_X_NODOUT
        const int64_t e = row;
#else
        const int64_t e = std::min<int64_t>((t0 + k) * TE + row, n_edges - 1);
#endif
        const float* src = dout + e * ld_dout + 8 * o;
        dr[0] = *reinterpret_cast<const v4f*>(src);
        dr[1] = *reinterpret_cast<const v4f*>(src + 4);
        dr[2] = *reinterpret_cast<const v4f*>(src + 64);
        dr[3] = *reinterpret_cast<const v4f*>(src + 68);
    };
    const int chunk_a = (o ^ (row & 15)) << 4, chunk_b = ((o + 8) ^ (row & 15)) << 4;
    auto load_members = [&](int k, v4f (&hm)[3]) {
        const int* idk = ids[k & 3] + row * 3;
#pragma unroll
#ifdef IHG_X_NOGATHER
        for (int m = 0; m < 3; ++m) hm[m] = *reinterpret_cast<const v4f*>(h + static_cast<int64_t>((idk[m] & 1) + row) * ld_h + QC * qtr + 4 * o);
#else
        for (int m = 0; m < 3; ++m) hm[m] = *reinterpret_cast<const v4f*>(h + static_cast<int64_t>(idk[m]) * ld_h + QC * qtr + 4 * o);
#endif
    };

    // UR: wave w scans columns 8 w .. 8 w + 7 of the quarter (lane & 7; the other lanes repeat them), every wave carries the same
    // (user, destination) state and its own columns' running sums from tile to tile
    int cur_user = -1, first_user = -1;
    float run_sum = 0.f;
    float* const first_slot = UR ? bnd_val + static_cast<int64_t>(2 * range) * D : nullptr;
    float* run_dst = first_slot;
    const int ucol = 8 * wave + (lane & 7), colg = QC * qtr + ucol;
    uint64_t walk_mask = 0;
    int walk_uid = 0;
    float carry_prev = 0.f;                                              // the running sum the scanned tile started from
    // straight-line part (sits in the matrix phase of the NEXT tile; for k < 0 it finds no rows): inclusive sums of the runs, row by
    // row, back into the image; rows past the end hold zeros
    auto scan_user_slot = [&](int k) {
        const int* idk = ids[k & 3];
        const int rows = k < 0 ? 0 : static_cast<int>(std::min<int64_t>(TE, n_edges - (t0 + k) * TE));
        const int r = std::max(lane < rows ? lane : rows - 1, 0);
        walk_uid = idk[r * 3];
        const int prev_uid = r == 0 ? cur_user : idk[(r - 1) * 3];
        walk_mask = __ballot(lane < rows && walk_uid != prev_uid);
        carry_prev = run_sum;
        float s = run_sum;
#pragma unroll
        for (int x0 = 0; x0 < TE; x0 += 8) {
            float v[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] = utile[x0 + x][ucol];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float keep = (walk_mask >> (x0 + x)) & 1 ? 0.f : 1.f;
                s = s * keep + v[x];
                utile[x0 + x][ucol] = s;
            }
        }
        run_sum = rows > 0 ? s : run_sum;
    };
    // the finished runs: one short iteration per run start in the tile
    auto emit_user_runs = [&]() {
        uint64_t m = walk_mask;
        while (m != 0) {
            const int x = __builtin_ctzll(m);
            m &= m - 1;
            const int user = __builtin_amdgcn_readlane(walk_uid, x);
            if (cur_user >= 0) {
                const float done = x == 0 ? carry_prev : utile[x - 1][ucol];     // the finished run: its sum up to the row before
                if (lane < 8) run_dst[colg] = done;
                run_dst = dh_user + static_cast<int64_t>(user) * ld_dh;
            } else {
                first_user = user;
            }
            cur_user = user;
        }
    };

    if (tid < 3 * TE) {
        ids[0][tid] = fetch_id(0);
        if (n_my > 1) ids[1][tid] = fetch_id(1);
    }
    v4f dr_a[4], dr_b[4];
    load_dout(0, dr_a);
    if (n_my > 1) load_dout(1, dr_b);
    {
        const Planes pa = split8(dr_a[0], dr_a[1]), pb = split8(dr_a[2], dr_a[3]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<v4u*>(&planes[0][p][row][chunk_a]) = pa.p[p];
            *reinterpret_cast<v4u*>(&planes[0][p][row][chunk_b]) = pb.p[p];
        }
    }
    __syncthreads();

    const int arow = lane & 15, kq = lane >> 4;
    // one tile: `use` holds the dout values of tile k + 1 (split here), `fill` is free and receives those of tile k + 2
    auto tile = [&](int k, v4f (&use)[4], v4f (&fill)[4]) {
        v4f hm[3];
        load_members(k, hm);
        if (k + 2 < n_my) load_dout(k + 2, fill);
        int id_next = 0;
        if (k + 2 < n_my && tid < 3 * TE) id_next = fetch_id(k + 2);
#ifndef IHG_X_NOSCAN
        if (UR) scan_user_slot(k - 1);
#endif

        v4f acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = v4f{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pbase = &planes[k & 1][0][0][0];
        unsigned char* pnext = &planes[(k + 1) & 1][0][row][0];
        auto fragment = [&](int step, v8s (&a)[3]) {
            const int kb = step >> 1, rt = step & 1;
            const unsigned char* src = pbase + (16 * rt + arow) * 256 + (((4 * kb + kq) ^ arow) << 4);
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const v8s*>(src + p * (TE * 256));
        };
        v8s a[3], an[3];
        fragment(0, a);
        v4u sp[3];
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int kb = step >> 1, rt = step & 1;
            if (step + 1 < 8) fragment(step + 1, an);
#ifndef IHG_X_NOSPLIT
            {   // two of the sixteen dout values of the next tile -> one dword of each plane
                const v4f x4 = use[step >> 1];
                const float xa = x4[2 * (step & 1)], xb = x4[2 * (step & 1) + 1];
                const float ra = xa - top16(xa), rb = xb - top16(xb);
                const float la = ra - top16(ra), lb = rb - top16(rb);
                sp[0][step & 3] = pack_hi(xa, xb);
                sp[1][step & 3] = pack_hi(ra, rb);
                sp[2][step & 3] = pack_hi(la, lb);
            }
#endif
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kTermA[term]], wreg[kb][ct][kTermB[term]], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if ((step & 3) == 3) {                                             // (past the last tile: nobody reads that image)
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<v4u*>(pnext + p * (TE * 256) + (step == 3 ? chunk_a : chunk_b)) = sp[p];
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = an[p];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) dzimg[wave][16 * rt + 4 * kq + r][16 * ct + arow] = acc[rt][ct][r];
#ifndef IHG_X_NOSCAN
        if (UR) emit_user_runs();
#endif
        __syncthreads();

        if (k + 2 < n_my && tid < 3 * TE) ids[(k + 2) & 3][tid] = id_next;
        {
            const int64_t e = (t0 + k) * TE + row;
            const v4f z_uq = *reinterpret_cast<const v4f*>(&dzimg[0][row][4 * o]), z_qi = *reinterpret_cast<const v4f*>(&dzimg[1][row][4 * o]);
            const v4f z_iu = *reinterpret_cast<const v4f*>(&dzimg[2][row][4 * o]), z_uqi = *reinterpret_cast<const v4f*>(&dzimg[3][row][4 * o]);
            const v4f hu = hm[0], hq = hm[1], hi = hm[2];
            const v4f g_u = z_uq * hq + z_iu * hi + z_uqi * (hq * hi);
            const v4f g_q = z_uq * hu + z_qi * hi + z_uqi * (hu * hi);
            const v4f g_i = z_qi * hq + z_iu * hu + z_uqi * (hu * hq);
            if (UR) *reinterpret_cast<v4f*>(&utile[row][4 * o]) = e < n_edges ? g_u : v4f{0.f, 0.f, 0.f, 0.f};
#ifdef IHG_X_NOSTORE
            if (e < n_edges && g_u[0] == 1234.5f) {
#else
            if (e < n_edges) {
#endif
                float* dst = g_out + e * ((UR ? 2 : 3) * D) + QC * qtr + 4 * o;
                if (!UR) {
                    store_stream4(dst, g_u);
                    dst += D;
                }
                store_stream4(dst, g_q);
                store_stream4(dst + D, g_i);
            }
        }
        __syncthreads();
    };
    for (int k = 0; k < n_my; k += 2) {
        tile(k, dr_b, dr_a);
        if (k + 1 < n_my) tile(k + 1, dr_a, dr_b);
    }
    if (UR) {
        scan_user_slot(n_my - 1);
        emit_user_runs();
        // the last run of the range may continue in the next one: second boundary slot - unless it IS the first run
        const bool one_run = run_dst == first_slot;
        if (cur_user >= 0 && lane < 8) {
            if (one_run) run_dst[colg] = run_sum;
            else bnd_val[static_cast<int64_t>(2 * range + 1) * D + colg] = run_sum;
        }
        if (qtr == 0 && tid == 0) {
            bnd_user[2 * range] = first_user;
            bnd_user[2 * range + 1] = (cur_user >= 0 && !one_run) ? cur_user : -1;
        }
    }
}

}  // namespace

int64_t split_plane_floats(int dim, int order) { return dim == 128 && order == 3 ? (3LL * 4 * dim * dim) / 2 : 0; }

bool split_arith_enabled() {
    static const bool enabled = [] {
        const char* v = std::getenv("IHG_INTERACT_ARITH");
        return v == nullptr || std::strcmp(v, "f32") != 0;
    }();
    return enabled;
}

bool split_members_ok(int dim, int order, const float* g, int64_t ld_h, int64_t ld_dout, const float* dout) {
    return split_arith_enabled() && dim == 128 && order == 3 && aligned16(g) && aligned16(dout) && ld_h % 4 == 0 && ld_dout % 4 == 0;
}

void launch_members_split(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, void* planes, const float* dout,
                          int64_t ld_dout, float* g, int64_t n_edges, float* dh_user, int64_t ld_dh, float* bnd_val, int32_t* bnd_user,
                          int* n_boundary_entries, hipStream_t s) {
    v4u* wsp = static_cast<v4u*>(planes);
    hipLaunchKernelGGL(pack_planes_members_kernel, dim3(4 * 4 * 4 * 2 * kWave / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, wsp);
    const int grid = 4 * kSplitRanges;
    if (dh_user != nullptr)
        hipLaunchKernelGGL(interact_bwd_members_split_kernel<true>, dim3(grid), dim3(kSplitThreads), 0, s, h, ld_h, i3, wsp, dout, ld_dout, g, n_edges, dh_user,
                           ld_dh, bnd_val, bnd_user);
    else
        hipLaunchKernelGGL(interact_bwd_members_split_kernel<false>, dim3(grid), dim3(kSplitThreads), 0, s, h, ld_h, i3, wsp, dout, ld_dout, g, n_edges,
                           static_cast<float*>(nullptr), int64_t{0}, static_cast<float*>(nullptr), static_cast<int32_t*>(nullptr));
    if (n_boundary_entries != nullptr) *n_boundary_entries = 2 * kSplitRanges;
}
