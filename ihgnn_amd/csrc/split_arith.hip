// split_arith.hip - contractions of the order-2 / order-3 interactive step PER HYPEREDGE (d = 64, 128, 256: member gradients, the hyperedge form's forward and
// weight gradients) on the 16-bit matrix pipe at fp32 accuracy.  The node-level contractions and linear maps are split_node.hip's; the helpers both share
// (operand splits, wave roles) split_common.hpp's.
//
// Every fp32 operand x is taken apart EXACTLY into three bf16 terms, x = hi + mid + lo (hi = the top 16 bits of x, mid = the top
// 16 bits of x - hi, lo = the rest: 8 + 8 + 8 significand bits, both subtractions exact), and a product a b is accumulated in fp32
// as the six partial products hi hi + hi mid + mid hi + hi lo + mid mid + lo hi, smallest first.  Each partial product of two
// bf16 values is exact in fp32.  The three that are left out (mid lo + lo mid + lo lo) are bounded by 2^-21 |a b|: the split truncates
// toward zero, so with the low 16 significand bits set mid ~ 2^-7 x and lo ~ 2^-15 x, and the omitted terms all carry the sign of the
// product they belong to - in a dot product whose terms share a sign they add up instead of averaging out (on random data the bias is
// ~ 2^-24).  Measured on exactly that worst case against float64 (tests/test_gpu_parity.py::test_split_arithmetic_worst_case_operands,
// every operand's low 16 bits set, all products positive), d = 64: forward 5.6e-7, weight gradients 5.2e-7, node-level map 4.5e-7, member
// gradients 1.3e-6 relative (two contractions deep: dout W, then the product rule in fp32); d = 256 (contraction length 1,024): forward
// 1.4e-6, member gradients 1.4e-6, node-level map 9.3e-7 - all at least 7 x inside the 1e-5 contract (the test holds 2e-6);
// on random data 2-4e-7, the fp32-MFMA kernels' own level (test_split_arithmetic_is_as_accurate_as_fp32_mfma).
// v_mfma_f32_16x16x32_bf16 runs 16 x the rate of v_mfma_f32_16x16x4_f32, six of them replace eight -> the same contraction in ~ 0.4 of
// the matrix-pipe time (tools/split_probe.hip has the standalone rate measurement).
//
// Three bf16 planes of the weights are 1.5 x their fp32 size: 384 KB at d = 128, more than one workgroup's registers can keep beside
// the accumulators.  So a workgroup owns a PART of the columns (a half at d = 128, an eighth at d = 256, all of them at d = 64); the
// parts of one tile range sit on one XCD (workgroups are dealt to XCDs round-robin by their linear id), so the later reads of a
// streamed row hit that L2.  The waves of a workgroup have two jobs, one of each per SIMD: matrix waves 0-3 only read fragments and
// issue MFMAs, service waves 4-7 request, split, apply the element-wise part and store.
// What bounds these kernels is instruction issue: a SIMD spends ~ 16 cycles per MFMA and ~ 4 per every other instruction of its matrix
// and its service wave, one after the other - so the count of instructions beside the MFMAs (the split: 5.5 per element; addresses;
// scalar bookkeeping) is what to cut; the pipe is 0.45-0.64 busy (DESIGN.md section 4 has the ladder, the probes and the counters).
#include "split_common.hpp"

namespace {

// planes of the member-gradient contraction dz_b[e][c] = sum_j dout[e][j] W[j][(3+b)d + c]   (k runs along j), two fp16 terms per weight:
// wsp[g][b][kb][ct][plane < 2][lane][8 x fp16] (g = 32-column group, d / 32 of them; kb < d / 32; ct < 2): element i = plane of
// wsc[b][c] W[32 kb + 8 (lane>>4) + i][(3+b) d + c],  c = 32 g + 16 ct + (lane&15);  wsc[b][c] = scale_up_for(max_j |W[j][(3+b)d + c]|)
// (dense_weight_scales_kernel over the four blocks as "types", transpose = 1)
__global__ __launch_bounds__(kBlockThreads) void pack_planes_members_kernel(const float* __restrict__ w, int64_t ld_w, int d, int nblk, const float* __restrict__ wsc,
                                                                            v4u* __restrict__ wsp) {
    const int kbs = d / 32, groups = d / 32;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= groups * 4 * kbs * 2 * kWave) return;
    const int lane = idx & 63, ct = (idx >> 6) & 1, kb = (idx >> 7) % kbs, b = ((idx >> 7) / kbs) & 3, g = (idx >> 7) / (kbs * 4);
    const int c = 32 * g + 16 * ct + (lane & 15);
    const float* src = w + static_cast<int64_t>(32 * kb + 8 * (lane >> 4)) * ld_w + (3 + b) * d + c;
    v4u hi = v4u{0u, 0u, 0u, 0u}, lo = hi;                               // (order 2 has three blocks: the fourth slot stays zero and is never read)
    if (b < nblk) {
        const float sc = wsc[b * d + c];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hh, ll;
            split_pair_h2(src[(2 * i) * ld_w] * sc, src[(2 * i + 1) * ld_w] * sc, hh, ll);
            hi[i] = hh;
            lo[i] = ll;
        }
    }
    wsp[(static_cast<int64_t>(idx >> 6) * 2 + 0) * kWave + lane] = hi;
    wsp[(static_cast<int64_t>(idx >> 6) * 2 + 1) * kWave + lane] = lo;
}

// Timeline probe of the member-gradient kernel (an ablation build's: -DIHG_ABL_M_TRACE; tools/phase_trace.py): lane 0 of matrix wave 0 and of service wave 4 of one
// workgroup stamp the clock at marks inside 1,024 phases; ihg_ablation_trace() copies the stamps out.
#ifdef IHG_ABL_M_TRACE
__device__ unsigned long long g_phase_trace[2][1024][4];
#define IHG_TRACE(role, k, mark)                                                                                   \
    if (blockIdx.x == 40 && lane == 0 && (k) >= 200 && (k) < 1224) g_phase_trace[role][(k) - 200][mark] = clock64();
#else
#define IHG_TRACE(role, k, mark)
#endif

// Member gradients dz_b = dout W_b, then the product rule.  A workgroup owns HALF of the columns (its weight planes: 128 KB) and its eight
// waves have two jobs, one of each per SIMD:
//   - waves 0-3, the matrix waves: wave = product block b, the half's 64 columns x the tile's 32 hyperedges, 96 MFMAs per tile (two fp16 terms per
//     operand, three products), weight planes resident (128 registers).  They read the dout fragments from two fp16 images (16-byte chunk o of row r
//     at o ^ (r & 15): conflict-free ds_read_b128) and issue the MFMA as W^T x dout^T, so that a lane holds 4 consecutive COLUMNS of one hyperedge;
//     the accumulators leave through the inverse scales (its columns' and the hyperedge's) into an LDS image in 16-byte pieces.
//   - waves 4-7, the service waves (256 threads): request rows two tiles ahead (dout: 16 values per thread; member values: 4 columns x 2 of
//     one hyperedge), scale (one power of two per hyperedge row: its eight threads agree on the largest magnitude) and split the next tile's dout
//     values into the images, apply the product rule to the previous tile (contractions from the image, member values from registers) and store.
//     Requested rows are taken delivery of (an opaque asm use) at the END of the phase
//     that requested them, before the barrier: the memory counter is in order, and left to the compiler the waits land in the next
//     phase's stream behind that phase's own requests and stores.
// An in-order wave that does both jobs stalls its MFMA stream on every wait of the service work (that form: 1,520 us, this one 1,370,
// same box, kbench scale).  Images, contraction image and id ring are double-buffered: one barrier per tile.  The two halves of a tile
// range sit on one XCD, so the second read of a dout row hits that L2.
// (Three bf16 terms per operand, six products, the user-slot sums on the service waves: 1,727 us at C3; two fp16 terms: 1,621; the sums moved
// to the matrix waves: 1,489; row maxima without canonicalisation and LDS shuffles: 1,404; the gathered rows requested a phase before the phase that sums them
// (two sets): - 4 % (1,404 - 1,470 inside a step, by box).  Ablation ladder of the 1,489 form: profiles/r4/09_abl_member_gradients_fp16.txt.)
// UR (hyperedges numbered by user; g is [E, 2, d]): the user-slot gradient is not stored per hyperedge.  The product rule leaves it in an
// LDS image [row][column]; a phase later MATRIX wave w adds up the image's rows 8 w .. 8 w + 7 (lane = column; a run = the rows of one
// user, its starts from one ballot over the tile's user ids) and stores the runs inside its window to dh[user] as 256-byte row pieces; a
// phase after that every wave closes the run that ends in its window from the carried sum and the windows' end pieces.  (user and -
// through LDS - the open run's sum) are carried from tile to tile, which is why a workgroup takes a CONTIGUOUS tile range; the first and the last run of a
// range may continue in the neighbours and go to the boundary table that interact.hip's user_boundary_fixup_kernel adds up (indexed by
// tile range here, shared by the two halves).  (First form: every wave scanned 16 columns of all 32 rows with a scalar reset factor and
// wrote the inclusive sums back for an emit loop - four times the additions on a quarter of the lanes, three LDS round trips in series.)
// D = 128: two column halves of 64 per tile range.  D = 256: eight parts of 32 columns (the weight planes are 1 MB), the dout tile is 32 KB
// and every part splits it again (UR there: round 5 - 32 lanes of a matrix wave carry the part's columns, the other 32 mirror them).  D = 64: one workgroup holds all of it.
// NBLK = 3 (order 2): matrix wave 3 has no block; it keeps the barriers and its window of the user sums.
// GATHER (D = 128, UR): there is no dout tensor yet - the cotangent of a hyperedge is the scaled sum of its three members' rows of a
// node-level cotangent dy ([N, d]; dout[e] = sum_m dy_scale[m] dy[m], the transpose of the hyperedge -> node pass that follows the
// interactive step in an IHGNN layer).  The service waves gather the three rows instead of streaming one, form the sum in the order of
// the node -> hyperedge kernel (K5) and store it to dout_store (each column half its 64 columns) for the weight-gradient kernel and the
// first-order scatter: K5's launch - memory-bound, on a chip whose issue slots it leaves idle - disappears into a kernel that is
// issue-bound and leaves the memory pipes idle.
// PLANES (round 5; D = 256): `dout` holds the hyperedges' cotangents already scaled and taken apart - rows of [plane][D] fp16, 4 D bytes like the fp32 row, written by
// edge_gather_sum_planes256_kernel (aggregate.hip) with the inverse scales in inv_src - so the service waves copy 16-byte pieces into the images instead of finding each
// row's maximum, scaling and splitting it in every one of the eight column parts (15 % of the kernel at config C5).
template <int D, bool UR, int NBLK, bool GATHER = false, bool PLANES = false>
__global__ __launch_bounds__(kSplitThreads) void interact_bwd_members_split_ws_kernel(const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3,
                                                                                      const v4u* __restrict__ wsp, const float* __restrict__ winv,
                                                                                      const float* __restrict__ dout, int64_t ld_dout,
                                                                                      float* __restrict__ g_out, int64_t n_edges, float* __restrict__ dh_user,
                                                                                      int64_t ld_dh, float* __restrict__ bnd_val, int32_t* __restrict__ bnd_user,
                                                                                      const float* __restrict__ dy_scale = nullptr,
                                                                                      float* __restrict__ dout_store = nullptr, int64_t ld_store = 0,
                                                                                      const float* __restrict__ inv_src = nullptr) {
    static_assert(D == 128 || D == 64 || D == 256, "shapes");
    static_assert(!PLANES || !GATHER, "the gathering form makes its cotangents itself");
    static_assert(!GATHER || ((D == 128 || D == 64) && UR), "the gathering form exists where the layer's backward uses it");
    constexpr int TE = kSplitTE, PARTS = D == 64 ? 1 : (D == 128 ? 2 : 8), RANGES = 256 / PARTS, HC = D / PARTS, CT = HC / 16, KB = D / 32, RB = 2 * D;
    constexpr int SWZ = RB / 16 - 1 < 15 ? RB / 16 - 1 : 15;            // the row swizzle stays inside a row (D = 64: rows of 8 chunks)
    constexpr int DOCT = D / 64, EX = HC / 32;                          // per service thread: dout octets, 4-column groups of the product rule
    constexpr int DZ = HC + 4, UTS = HC + 4, GS = UR ? 2 : 3;
    __shared__ __attribute__((aligned(16))) unsigned char planes[2][2][TE][RB];
    __shared__ float sinv[2][TE];                                        // inverse scales of the rows whose images are in planes[.]
    __shared__ __attribute__((aligned(16))) float dzimg[2][4][TE][DZ];
    __shared__ __attribute__((aligned(16))) float utile[UR ? 2 : 1][UR ? TE : 1][UTS];    // user-slot gradients of a tile, [row][column of the half]
    __shared__ float ucarry[2][UR ? HC : 1];                              // sum so far of the run that is open when tile t begins: [t & 1]
    __shared__ float uhead[2][4][UR ? HC : 1], utail[2][4][UR ? HC : 1];  // per 8-row window of a tile: sum before its first run start / after its last
    __shared__ int ids[8][3 * TE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int half = (bid >> 3) & (PARTS - 1), range = (bid & 7) + 8 * (bid / (8 * PARTS));      // `half`: this workgroup's column part
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t per = (n_tiles + RANGES - 1) / RANGES;
    const int64_t t0 = range * per;
    const int n_my = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(per, n_tiles - t0)));
    if (n_my == 0) {
        if (UR && half == 0 && tid == 0) bnd_user[2 * range] = bnd_user[2 * range + 1] = -1;
        return;
    }
    const int n_phases = n_my + (UR ? 3 : 1);

    // UR, lane = column of the half (64 lanes, 256-byte stores); a run = the rows of one user, its starts come from one ballot over the
    // tile's user ids.  The sums ride on the MATRIX waves: with two fp16 terms a tile's contraction keeps the matrix pipe busy for a quarter of
    // a phase, the service waves are the ones on the kernel's critical path (without the sums: - 320 us of 1,900, profiles/r4/01_abl_member_gradients.txt),
    // and this work is wave-shaped anyway (four 8-row windows, lane = column):
    //   tile t, phase t + 2: matrix wave w forms the running sums of rows 8 w .. 8 w + 7 (read at the START of the phase; a run start
    //     resets the sum through a scalar factor: two instructions per row, no branch), leaves the sum before the window's first run
    //     start and the one after its last in LDS and stores the runs that lie inside the window (a loop over the window's run starts);
    //   phase t + 3: wave w closes the run that ends at ITS window's first run start: carried sum (when no run started earlier in the
    //     tile) + the end sums of the windows since + its own start sum, added in that order; wave 3 leaves the new carried sum.
    // Every wave keeps the open run's user and whether it still is the range's first run.
    int cur_user = -1, first_user = -1;
    bool first_run_open = true;                                      // no run has ended yet in this range
    float* const first_slot = UR ? bnd_val + static_cast<int64_t>(2 * range) * D : nullptr;
    // (D = 256: parts of 32 columns - the upper half of a wave mirrors the lower one's column (ul) through the sums and leaves the stores to it (uw))
    const int ul = lane & (HC - 1);
    const bool uw = lane < HC;
    const int colg = HC * half + ul, win = wave & 3;
    // D = 256 (round 5): the second step - closing the runs that cross windows - is the SERVICE waves' (window = wave & 3 there too; they carry the open run's user
    // and rebuild a tile's run starts from the id ring with the same ballot).  A clock probe inside the kernel (tools/phase_trace.py) showed a phase of 4,050 cycles with the
    // matrix waves at its end: 2,790 for the 96 MFMAs (their fragments wait for an LDS pipe that the eight parts' re-reads of the cotangent tile keep busy) + 970 for the two
    // steps of the sums, the service waves done after 3,010.  (All of the sums on the service waves: C5 332.9 against 329.3 ms.)
    constexpr bool CHAIN_SVC = UR && D == 256;
    uint64_t heads_prev = 0;                                         // run starts / user ids / rows of the tile whose windows were summed a phase ago
    int uid_prev = 0, rows_prev = 0;
    auto tile_heads = [&](int t, int last_user, int& uid, int& rows) {   // run starts of tile t (bit r: row r begins a run), given the user of the row in front of it
        const int* idk = ids[t & 7];
        rows = static_cast<int>(std::min<int64_t>(TE, n_edges - (t0 + t) * TE));
        const int r = lane < rows ? lane : rows - 1;
        uid = idk[r * 3];
        const int prev_uid = r == 0 ? last_user : idk[(r - 1) * 3];
        return __ballot(lane < rows && uid != prev_uid);
    };
    auto run_target = [&](int t, int r, int uid) {                   // destination of the run that starts at row r of tile t
        return (t == 0 && r == 0) ? first_slot : dh_user + static_cast<int64_t>(__builtin_amdgcn_readlane(uid, r)) * ld_dh;
    };
    auto load_window = [&](int t, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = utile[t & 1][8 * win + i][ul];
    };
    auto sum_window = [&](int t, const float (&v)[8]) {              // tile t, 0 <= t < n_my
        int uid, rows;
        // (CHAIN_SVC: this role does not carry the open run's user - the last row of the tile before, full by construction, is still in the id ring)
        const uint64_t m = tile_heads(t, CHAIN_SVC ? (t == 0 ? -1 : ids[(t - 1) & 7][(TE - 1) * 3]) : cur_user, uid, rows);
        heads_prev = m;
        uid_prev = uid;
        rows_prev = rows;
        const unsigned mw = static_cast<unsigned>(m >> (8 * win)) & 0xffu;
        float pre[8], sum = 0.f;                                     // (rows past the end hold zeros and start no run)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sum = sum * ((mw >> i) & 1 ? 0.f : 1.f) + v[i];
            pre[i] = sum;
        }
        utail[t & 1][win][ul] = sum;
        if (mw != 0) {                                               // (0.85 run starts per window on C3)
            // Row i + 1 starts a run: the run row i belongs to ends at row i.  Begun inside the window (a start at or before i): its sum is pre[i] and goes to its user's
            // row; begun earlier: pre[i] is the window's piece in front of its first start (uhead; a start at the window's row 0: nothing in front).  Unrolled over i with
            // wave-uniform conditions - compile-time register indices, no selection chains.  (Round 4's form looped over the set bits and picked pre[next - 1] through seven
            // selects per run: 310 instructions and 34 branches in a matrix wave's phase beside its 96 MFMAs.)
            if (mw & 1u) uhead[t & 1][win][ul] = 0.f;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const unsigned upto = mw & ((2u << i) - 1u);         // starts at window rows 0 .. i
                if ((mw >> (i + 1)) & 1u) {
                    if (upto != 0) {
                        float* dst = (t == 0 && win == 0 && upto == 1u) ? first_slot
                                                                         : dh_user + static_cast<int64_t>(__builtin_amdgcn_readlane(uid, 8 * win + i)) * ld_dh;
                        if (uw) dst[colg] = pre[i];
                    } else {
                        uhead[t & 1][win][ul] = pre[i];
                    }
                }
            }
        }
    };
    // (Round 5, D = 256: a "run plan" - the service waves work out a tile's run starts, reset factors and destination addresses a phase ahead (lane = row) and this step reads
    // them instead of rebuilding them with scalar instructions: the two steps 980 -> 550 cycles, the MFMA loop 2,680 -> 3,140, the phase 3,880 -> 4,060: the SIMD's issue port
    // is shared, what the plan's LDS traffic and the service waves' extra instructions cost lands on the same phase.  Removed.)
    // (Round 5, D = 256, tools/phase_trace.py: this step costs 830 of a phase's 3,900 cycles behind the MFMA loop.  Cut into ten pieces riding INSIDE the loop - one per k-step
    // between the request of the next step's fragments and the first MFMA of this one - the loop grew from 2,680 to 3,830 cycles: its lost cycles are not waits an in-order wave
    // could fill.  With the id reads sent out at the phase's start and the rest behind the loop: 2,910 + 860.  Both removed.)
    struct Chain {
        float tail[4], head, carry;
    };
    auto load_chain = [&](int t, Chain& c) {
#pragma unroll
        for (int w = 0; w < 4; ++w) c.tail[w] = utail[t & 1][w][ul];
        c.head = uhead[t & 1][win][ul];
        c.carry = ucarry[t & 1][ul];
    };
    auto chain_windows = [&](int t, const Chain& c) {                // tile t, a phase after sum_window(t)
        if (CHAIN_SVC) heads_prev = tile_heads(t, cur_user, uid_prev, rows_prev);
        const unsigned m = static_cast<unsigned>(heads_prev);
        const unsigned before_me = m & ((1u << (8 * win)) - 1u);     // run starts in earlier windows of the tile
        const int p = before_me != 0 ? (31 - __builtin_clz(before_me)) >> 3 : -1;            // the last earlier window that has one
        if (((m >> (8 * win)) & 0xffu) != 0 || win == 3) {
            float sum = p < 0 ? c.carry : 0.f;
#pragma unroll
            for (int w = 0; w < 3; ++w)
                if (w < win && w >= p) sum += c.tail[w];
            if (((m >> (8 * win)) & 0xffu) != 0) {                   // the open run ends at this window's first run start
                float* dst = nullptr;
                if (p >= 0) dst = run_target(t, 31 - __builtin_clz(before_me), uid_prev);
                else if (cur_user >= 0) dst = first_run_open ? first_slot : dh_user + static_cast<int64_t>(cur_user) * ld_dh;
                if (dst != nullptr && uw) dst[colg] = sum + c.head;
                sum = 0.f;
            }
            if (win == 3) ucarry[(t + 1) & 1][ul] = sum + c.tail[3];
        }
        if (t == 0) first_user = __builtin_amdgcn_readlane(uid_prev, 0);
        if (m != 0) first_run_open = t == 0 && m == 1u;
        cur_user = __builtin_amdgcn_readlane(uid_prev, rows_prev - 1);
    };
    auto close_range = [&]() {                                       // after the last phase, by the role that closes the runs
        // the last run of the range may continue in the next one: second boundary slot - unless it IS the first run
        const bool one_run = first_run_open;
        if (cur_user >= 0 && win == 0 && uw) {
            const float run_sum = ucarry[n_my & 1][ul];              // (written before the last barrier)
            if (one_run) first_slot[colg] = run_sum;
            else bnd_val[static_cast<int64_t>(2 * range + 1) * D + colg] = run_sum;
        }
        if (half == 0 && win == 0 && lane == 0) {
            bnd_user[2 * range] = first_user;
            bnd_user[2 * range + 1] = (cur_user >= 0 && !one_run) ? cur_user : -1;
        }
    };
    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- service waves: 256 threads, thread -> hyperedge row, dout octets o and o + 8, member / gradient columns 4 o .. and 32 + 4 o ..
        const int st = tid - 256, row = st >> 3, o = st & 7;
        constexpr int AHEAD = GATHER ? 1 : 0;                            // the gathering form requests its rows a tile further ahead: its ids travel a phase earlier
        const int64_t last_pos = n_edges * 3 - 1;
        const uint32_t ldh = static_cast<uint32_t>(ld_h), ldd = static_cast<uint32_t>(ld_dout);
        // tile bases and the clamps at the end of the hyperedge list are scalar (the tile number is uniform); per lane: one min, one multiply
        auto fetch_id = [&](int k) {                                     // (st < 96; the tile exists)
            const int64_t first = (t0 + k) * (3 * TE);
            const int lim = static_cast<int>(std::min<int64_t>(last_pos - first, 3 * TE - 1));
            return (i3 + first)[std::min(st, lim)];
        };
        auto load_dout = [&](int k, v4f (&dr)[2 * DOCT], float& iv) {    // (tiles past the end: the last row, dropped)
            const int64_t first = std::min<int64_t>((t0 + k) * TE, n_edges - 1);
            const int lim = static_cast<int>(std::min<int64_t>(n_edges - 1 - first, TE - 1));
            if (PLANES) {                                                // 16-byte piece o + 8 x of either plane, as it goes into the image; the row's inverse scale first (the oldest request)
                if (abl::m_no_dy_loads) {
                    iv = 1.f;
#pragma unroll
                    for (int x = 0; x < 2 * DOCT; ++x) dr[x] = v4f{1.f, 2.f, 3.f, 4.f} * static_cast<float>(lim + x);
                    return;
                }
                iv = (inv_src + first)[std::min(row, lim)];
                const float* src = row_at(dout + first * ld_dout, std::min(row, lim), ldd) + 4 * o;
#pragma unroll
                for (int x = 0; x < DOCT; ++x) {
                    dr[2 * x] = *reinterpret_cast<const v4f*>(src + 32 * x);
                    dr[2 * x + 1] = *reinterpret_cast<const v4f*>(src + D / 2 + 32 * x);
                }
                return;
            }
            const float* src = row_at(dout + first * ld_dout, std::min(row, lim), ldd) + 8 * o;
#pragma unroll
            for (int x = 0; x < DOCT; ++x) {
                if (abl::m_no_dy_loads) {
                    dr[2 * x] = dr[2 * x + 1] = v4f{1.f, 2.f, 3.f, 4.f} * static_cast<float>(lim + x);
                    continue;
                }
                dr[2 * x] = *reinterpret_cast<const v4f*>(src + 64 * x);
                dr[2 * x + 1] = *reinterpret_cast<const v4f*>(src + 64 * x + 4);
            }
        };
        // GATHER: `dout` is the node-level cotangent dy; rows of the three members of hyperedge `row` of tile k, and their scales
        struct Raw {
            v4f r[3][2 * DOCT];
            float s[3];
        };
        auto load_gather = [&](int k, Raw& raw) {
            const int* idk = ids[k & 7] + row * 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int id = idk[m];
                const float* src = row_at(dout, id, ldd) + 8 * o;
#pragma unroll
                for (int x = 0; x < DOCT; ++x) {
                    if (abl::m_no_dy_loads) {
                        raw.r[m][2 * x] = raw.r[m][2 * x + 1] = v4f{1.f, 2.f, 3.f, 4.f} * static_cast<float>(id + x);
                        continue;
                    }
                    raw.r[m][2 * x] = *reinterpret_cast<const v4f*>(src + 64 * x);
                    raw.r[m][2 * x + 1] = *reinterpret_cast<const v4f*>(src + 64 * x + 4);
                }
                raw.s[m] = abl::m_no_dy_loads ? 0.5f : (dy_scale != nullptr ? dy_scale[id] : 1.f);
            }
        };
        auto combine = [&](int k, const Raw& raw, v4f (&dr)[2 * DOCT]) {   // K5's order: ((0 + s_u u) + s_q q) + s_i i; the half's columns go to dout_store
#pragma unroll
            for (int j = 0; j < 2 * DOCT; ++j) {
                v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 3; ++m) acc += raw.s[m] * raw.r[m][j];
                dr[j] = acc;
            }
            const int64_t e0 = (t0 + k) * TE;
            if (!abl::m_no_dout_store && ld_store > 0 && k < n_my && e0 + row < n_edges) {       // (ld_store <= 0: nobody reads the hyperedges' cotangents after this kernel)
                float* dst = dout_store + (e0 + row) * ld_store + 64 * half + 8 * o;
                store_stream4(dst, dr[2 * half]);
                store_stream4(dst + 4, dr[2 * half + 1]);
            }
        };
        auto load_members = [&](int k, v4f (&hm)[EX][3]) {
            const int* idk = ids[k & 7] + row * 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float* hp = row_at(h, idk[m], ldh) + HC * half + 4 * o;
#pragma unroll
                for (int x = 0; x < EX; ++x) hm[x][m] = abl::m_no_member_loads ? v4f{1.f, 2.f, 3.f, 4.f} * static_cast<float>(idk[m] + x) : *reinterpret_cast<const v4f*>(hp + 32 * x);
            }
        };
        const int swz = row & SWZ;
        // a row's cotangent goes in scaled by ONE power of two: its eight staging threads (consecutive lanes) agree on its largest magnitude with three shuffles
        auto split_tile = [&](const v4f (&dr)[2 * DOCT], int buf, float iv) {
            if (PLANES) {
#pragma unroll
                for (int x = 0; x < DOCT; ++x) {
                    *reinterpret_cast<v4f*>(&planes[buf][0][row][((o + 8 * x) ^ swz) << 4]) = dr[2 * x];
                    *reinterpret_cast<v4f*>(&planes[buf][1][row][((o + 8 * x) ^ swz) << 4]) = dr[2 * x + 1];
                }
                if (o == 0) sinv[buf][row] = iv;
                return;
            }
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * DOCT; ++j) m = abs_max3(dr[j][2], dr[j][3], abs_max3(dr[j][0], dr[j][1], m));
            m = row_lanes_max<8>(m);
            float inv;
            const float sc = abl::m_no_split ? 1.f : scale_up_for(m, inv);
#pragma unroll
            for (int x = 0; x < DOCT; ++x) {
                v4u hi, lo;
                if (abl::m_no_split) {
                    hi = __builtin_bit_cast(v4u, dr[2 * x]);
                    lo = __builtin_bit_cast(v4u, dr[2 * x + 1]);
                } else {
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        unsigned hh, ll;
                        split_pair_h2(dr[2 * x + (pr >> 1)][2 * (pr & 1)] * sc, dr[2 * x + (pr >> 1)][2 * (pr & 1) + 1] * sc, hh, ll);
                        hi[pr] = hh;
                        lo[pr] = ll;
                    }
                }
                *reinterpret_cast<v4u*>(&planes[buf][0][row][((o + 8 * x) ^ swz) << 4]) = hi;
                *reinterpret_cast<v4u*>(&planes[buf][1][row][((o + 8 * x) ^ swz) << 4]) = lo;
            }
            if (o == 0) sinv[buf][row] = abl::m_no_split ? 1.f : inv;
        };
        // product rule of tile k - 1 (its contractions in dzimg[(k - 1) & 1], its member values in hm) and the stores
        auto epilogue = [&](int k, const v4f (&hm)[EX][3]) {
            const float (*dz)[TE][DZ] = dzimg[(k - 1) & 1];
            const int64_t e = (t0 + k - 1) * TE + row;
#pragma unroll
            for (int x = 0; x < EX; ++x) {
                const int c = 4 * o + 32 * x;
                const v4f z_uq = *reinterpret_cast<const v4f*>(&dz[0][row][c]), z_qi = *reinterpret_cast<const v4f*>(&dz[1][row][c]);
                const v4f z_iu = *reinterpret_cast<const v4f*>(&dz[2][row][c]);
                const v4f hu = hm[x][0], hq = hm[x][1], hi = hm[x][2];
                // (round 5: the same rule element by element - no packed fp32 instructions beside the matrix waves' MFMAs - measured the same: phase 3,900 against 3,880 cycles)
                v4f g_u = z_uq * hq + z_iu * hi, g_q = z_uq * hu + z_qi * hi, g_i = z_qi * hq + z_iu * hu;
                if (NBLK == 4) {
                    const v4f z_uqi = *reinterpret_cast<const v4f*>(&dz[3][row][c]);
                    g_u += z_uqi * (hq * hi);
                    g_q += z_uqi * (hu * hi);
                    g_i += z_uqi * (hu * hq);
                }
                if (abl::m_epi_nomath) {                                 // the image reads and the stores, no arithmetic
                    g_u = z_iu;
                    g_q = z_uq;
                    g_i = z_qi;
                }
                const bool live = e < n_edges;
                if (UR) *reinterpret_cast<v4f*>(&utile[(k - 1) & 1][row][c]) = live ? g_u : v4f{0.f, 0.f, 0.f, 0.f};     // (rows past the end: zeros for the sums)
                if ((live || abl::m_uncond) && !abl::m_no_g_stores) {
                    // (where the stores land does not matter: into a 48 MB window, as one contiguous stream per column part or as plain stores the kernel takes the same
                    // time - ablate.hpp, profiles/r5/22_abl_member_stores_d256.txt: removing them "saves" 10 of 18 ms only because the product rule goes with them)
                    const int64_t er = abl::m_g_window ? (e & 0x3fff) : e;
                    constexpr int SS = abl::m_g_parts ? HC : D;
                    float* dst = abl::m_g_parts ? g_out + (static_cast<int64_t>(half) * n_edges + er) * (GS * HC) + c : g_out + er * (GS * D) + HC * half + c;
                    auto put = [](float* p, v4f v) {
                        if (abl::m_g_plain) *reinterpret_cast<v4f*>(p) = v;
                        else store_stream4(p, v);
                    };
                    if (!UR) {
                        put(dst, g_u);
                        dst += SS;
                    }
                    put(dst, g_q);
                    put(dst + SS, g_i);
                }
            }
        };
        if (UR && st < HC) ucarry[0][st] = 0.f;
        if (st < 3 * TE) {
            ids[0][st] = fetch_id(0);
#pragma unroll
            for (int k = 1; k < 8; ++k) ids[k][st] = k < 4 + AHEAD && k < n_my ? fetch_id(k) : 0;       // (slots of tiles past the end: row 0, requested and dropped)
        }
        __syncthreads();
        v4f dr0[2 * DOCT], dr1[2 * DOCT], hm0[EX][3], hm1[EX][3];                        // dout values of tile m in dr<m & 1>, member values in hm<m & 1>
        // (GATHER) the rows behind the dout values: TWO sets - the rows of tile k + 3 are requested in phase k and summed at the end of phase k + 1 (one set, requested and
        // summed inside one phase, made every phase wait out a loaded memory round trip)
        Raw raw0, raw1;
        float iv0 = 1.f, iv1 = 1.f;                                      // (PLANES) inverse row scales beside dr0 / dr1
        if (GATHER) {                                                    // one set of dout values: a tile's sum is formed after the previous one was split
            load_gather(0, raw0);
            combine(0, raw0, dr0);
            split_tile(dr0, 0, 1.f);
            load_gather(1, raw0);
            load_gather(2, raw1);
            combine(1, raw0, dr0);
        } else {
            load_dout(0, dr0, iv0);
            if (n_my > 1) load_dout(1, dr1, iv1);
            split_tile(dr0, 0, iv0);
        }
        __syncthreads();
        int id_carry = 0;
        auto phase = [&](int k, v4f (&use)[2 * DOCT], v4f (&fill)[2 * DOCT], v4f (&hm_cur)[EX][3], v4f (&hm_prev)[EX][3], Raw& raw_use, Raw& raw_req, float& iv_use,
                         float& iv_fill) {
            // (Round 5 tried the streaming form - config C5's kernel - with this phase's requests first and then the work on what was requested a PHASE AGO behind one exact
            // wait, every request unconditional: the loop's waits became vmcnt(11 .. 22) instead of two vmcnt(0) per trip, and the kernel went 29.7 -> 31.1 ms at C5: it does not
            // wait for latency - 78 GB of L2-miss traffic and a matrix pipe half busy share its 30 ms.  Reverted.)
            // The id fetch of the tile four ahead comes FIRST and is unconditional (every lane, clamped tile - round 5): it used to sit behind the row requests under
            // `k + 4 + AHEAD < n_my && st < 96`, the only requests younger than it were the epilogue's stores - themselves under `live` - and the wait for it at the top of the
            // next phase was vmcnt(0): every other phase drained the queue, the previous phase's six stores included.  With the row requests behind it the wait is an exact
            // count that leaves them (and the stores) in flight.
            if (wave == 4) { IHG_TRACE(1, k, 0) }
            const int id_new = fetch_id(std::min(k + 4 + AHEAD, n_my - 1));
            Chain chain;
            if (CHAIN_SVC && !abl::m_no_user_sums && k >= 3) load_chain(k - 3, chain);      // LDS reads of the run sums this phase closes, used at its end
            // ids of tile k + 3 + AHEAD (requested in the previous phase) into the ring: a phase ahead of their first readers (the gathered rows of
            // that tile in the next phase)
            if (k >= 1 && k + 3 + AHEAD < n_my && st < 3 * TE) ids[(k + 3 + AHEAD) & 7][st] = id_carry;
            load_members(k, hm_cur);                                     // unconditional: a branch around requests costs whole-set register copies
            if (GATHER) load_gather(k + 3, raw_req);
            else load_dout(k + 2, fill, iv_fill);
            id_carry = id_new;
            if (wave == 4) { IHG_TRACE(1, k, 1) }
            if (k + 1 < n_my) split_tile(use, (k + 1) & 1, iv_use);
            if (wave == 4) { IHG_TRACE(1, k, 2) }
            // delivery of this phase's requests, THEN everything that stores: the memory counter is in order, a wait behind a store sits out
            // the store's round trip to memory
            if (GATHER) {                                                // (the rows requested a phase ago: this phase's requests stay in flight)
                asm volatile("" : "+v"(raw_use.r[2][2 * DOCT - 2]), "+v"(raw_use.r[2][2 * DOCT - 1]), "+v"(raw_use.s[2]));
            } else if (!abl::m_uncond) {
                asm volatile("" : "+v"(fill[0]), "+v"(fill[1]), "+v"(fill[2 * DOCT - 2]), "+v"(fill[2 * DOCT - 1]));     // (in order: the last delivered = all delivered)
            }
            if (!abl::m_uncond) asm volatile("" : "+v"(hm_cur[EX - 1][0]), "+v"(hm_cur[EX - 1][1]), "+v"(hm_cur[EX - 1][2]));       // (left to where the next phase reads them: no gain)
            if (GATHER) combine(k + 2, raw_use, fill);
            if (abl::m_uncond) epilogue(std::min(std::max(k, 1), n_my), hm_prev);
            else if (k >= 1 && k - 1 < n_my && !abl::m_no_product_rule) epilogue(k, hm_prev);
            if (CHAIN_SVC && !abl::m_no_user_sums && k >= 3) chain_windows(k - 3, chain);
            if (wave == 4) { IHG_TRACE(1, k, 3) }
            __syncthreads();
        };
        int k = 0;
#pragma clang loop unroll(disable)
        for (; k + 1 < n_phases; k += 2) {                                // exactly two phases per trip: the register sets come back in place
            phase(k, GATHER ? dr0 : dr1, dr0, hm0, hm1, raw1, raw0, iv1, iv0);
            phase(k + 1, dr0, GATHER ? dr0 : dr1, hm1, hm0, raw0, raw1, iv0, iv1);
        }
        if (k < n_phases) phase(k, GATHER ? dr0 : dr1, dr0, hm0, hm1, raw1, raw0, iv1, iv0);
        if (CHAIN_SVC) close_range();
        return;
    }

    // ---------------- matrix waves: wave = product block, the half's four 16-column tiles, both row tiles
    const int blk = wave;
    const int arow = lane & 15, kq = lane >> 4;
    v8h wreg[KB][CT][2];
    v4f wiv[CT];                                                         // inverse scales of this lane's output columns 16 ct + 4 kq ..
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int p = 0; p < 2; ++p)      // 32-column group of column tile ct: (HC / 32) half + (ct >> 1)
                wreg[kb][ct][p] = __builtin_bit_cast(v8h, wsp[(static_cast<int64_t>((((HC / 32) * half + (ct >> 1)) * 4 + blk) * (KB * 2) + kb * 2 + (ct & 1)) * 2 + p) * kWave + lane]);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wiv[ct] = blk < NBLK ? *reinterpret_cast<const v4f*>(winv + blk * D + HC * half + 16 * ct + 4 * kq) : v4f{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    __syncthreads();
    for (int k = 0; k < n_phases; ++k) {
        float wrows[8];
        Chain chain;
        if (wave == 0) { IHG_TRACE(0, k, 0) }
        const bool sums_now = UR && !abl::m_no_user_sums && k >= 2 && k - 2 < n_my;
        if (UR && !abl::m_no_user_sums) {                                // LDS reads of this phase's run sums, used at its end
            if (sums_now) load_window(k - 2, wrows);
            if (!CHAIN_SVC && k >= 3) load_chain(k - 3, chain);
        }
        const bool mfma_now = k < n_my && blk < NBLK && !abl::m_no_mfma;
        if (mfma_now) {
            const unsigned char* pbase = &planes[k & 1][0][0][0];
            // row tile after row tile (one set of CT accumulator tiles live), the fragments of step s + 1 requested in front of the MFMAs of step s.  (D = 256, round 5, clock
            // probe: the loop takes 2,680 cycles for 96 MFMAs = 1,536 cycles of matrix pipe.  Fragments two steps ahead: 2,710.  Both row tiles at once - four accumulator
            // chains instead of two: 2,640.  Neither the LDS latency nor the accumulator dependency is what it loses; the same loop beside a service role that only copies: see DESIGN.)
            auto fragment = [&](int step, v8h (&a)[2]) {
                const int rt = step / KB, kb = step % KB;
                const unsigned char* src = pbase + (16 * rt + arow) * RB + (((4 * kb + kq) ^ (arow & SWZ)) << 4);
#pragma unroll
                for (int p = 0; p < 2; ++p) a[p] = *reinterpret_cast<const v8h*>(src + p * (TE * RB));
            };
            const float iv[2] = {sinv[k & 1][arow], sinv[k & 1][16 + arow]};
            v8h a[2], an[2];
            v4f acc[CT];
            fragment(0, a);
#pragma unroll
            for (int step = 0; step < 2 * KB; ++step) {
                const int rt = step / KB, kb = step % KB;
                if (kb == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = v4f{0.f, 0.f, 0.f, 0.f};
                }
                if (step + 1 < 2 * KB) fragment(step + 1, an);
                IHG_PIN_ORDER();                                         // keep the next step's reads IN FRONT of this step's MFMAs
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[kb][ct][kTermB2[term]], a[kTermA2[term]], acc[ct], 0, 0, 0);
                IHG_PIN_ORDER();
#pragma unroll
                for (int p = 0; p < 2; ++p) a[p] = an[p];
                if (kb == KB - 1) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) *reinterpret_cast<v4f*>(&dzimg[k & 1][blk][16 * rt + arow][16 * ct + 4 * kq]) = acc[ct] * (wiv[ct] * iv[rt]);
                }
            }
        }
        if (wave == 0) { IHG_TRACE(0, k, 1) }
        if (UR && !abl::m_no_user_sums) {
            if (!CHAIN_SVC && k >= 3) chain_windows(k - 3, chain);       // (before the next tile's ballot: it compares with the open run's user)
            if (sums_now) sum_window(k - 2, wrows);
        }
        if (wave == 0) { IHG_TRACE(0, k, 2) }
        __syncthreads();
    }
    if (UR && !CHAIN_SVC) close_range();
}

// ------------------------------------------------------------------------------------------------
// Forward: out[e][j] = sum_b sum_c z_b[e][c] W_b[j][c] + (P[u] + P[q]) + P[i].  A workgroup owns HALF of the output columns (its weight
// planes: 192 KB), so both halves form and split every product: the forward carries twice the split work of the gradient kernels.
// Matrix wave b (waves 0-3, one per SIMD) contracts product block b - 128 values of the contraction index, weight planes for the half's
// 64 output columns in 192 registers, 96 MFMAs per 16-hyperedge tile, issued as W x z^T so that a lane holds 4 consecutive output columns
// of one hyperedge - from bf16 images of z that the service waves 4-7 lay down a tile ahead: each of their 256 threads gathers 8 columns of
// one hyperedge's three member rows straight into registers (two tiles ahead), forms the 32 products, splits them and writes 8-byte
// pieces (chunk c of row r at c ^ r, rows of 1 KB: conflict-free ds_read_b128 for the fragments).  The four partial sums of a tile meet
// in an LDS image; the service waves add them in block order with the first-order rows (requested a tile earlier) and store, a phase
// later.  Images, partial-sum image and id ring are double-buffered: one barrier per tile.  (Every wave doing both jobs, members staged
// by LDS-DMA: 2,215 us against 2,020 on the same box; alone, the matrix waves would take 1,350 us and the service waves 1,380.)
// ------------------------------------------------------------------------------------------------
constexpr int kFwdTE = 16;

// wsp[half][b][jt][kb][plane][lane][8] (half < d / 64, kb < d / 32): element i = plane of W[64 half + 16 jt + (lane & 15)][(3 + b) d + 32 kb + 8 (lane >> 4) + i]
__global__ __launch_bounds__(kBlockThreads) void pack_planes_fwd_kernel(const float* __restrict__ w, int64_t ld_w, int d, int nblk, v4u* __restrict__ wsp) {
    const int kbs = d / 32, halves = d / 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= halves * 4 * 4 * kbs * kWave) return;
    const int lane = idx & 63, kb = (idx >> 6) % kbs, jt = ((idx >> 6) / kbs) & 3, b = ((idx >> 6) / (kbs * 4)) & 3, half = (idx >> 6) / (kbs * 16);
    const float* src = w + static_cast<int64_t>(64 * half + 16 * jt + (lane & 15)) * ld_w + (3 + b) * d + 32 * kb + 8 * (lane >> 4);
    const Planes pl = b < nblk ? split8(v4f{src[0], src[1], src[2], src[3]}, v4f{src[4], src[5], src[6], src[7]})
                               : split8(v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int p = 0; p < 3; ++p) wsp[(static_cast<int64_t>(idx >> 6) * 3 + p) * kWave + lane] = pl.p[p];
}

// D = 128: two output-column halves per tile sequence (both form and split every product).  D = 64: one workgroup, no duplicated work.
// NBLK = 3 (order 2): matrix wave 3 has no block and only keeps the barriers.
template <int D, int NBLK>
__global__ __launch_bounds__(kSplitThreads) void interact_fwd_split_ws_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p,
                                                                              const int32_t* __restrict__ i3, const v4u* __restrict__ wsp, float* __restrict__ out,
                                                                              int64_t ld_out, int64_t n_edges) {
    constexpr int TE = kFwdTE, PARTS = D / 64, RANGES = 256 / PARTS, HC = 64, PS = HC + 4, KB = D / 32, ZX = D / 64;
    constexpr int ZRB = 2 * 4 * D;                                       // bytes of one hyperedge's row of a z image (4 blocks x D columns)
    constexpr int ZPL = TE * ZRB;
    __shared__ __attribute__((aligned(16))) unsigned char zplanes[2][3][TE][ZRB];
    __shared__ __attribute__((aligned(16))) float part[2][4][TE][PS];
    __shared__ int ids[8][3 * TE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int half = (bid >> 3) & (PARTS - 1), range = (bid & 7) + 8 * (bid / (8 * PARTS));
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int n_my = range < n_tiles ? static_cast<int>((n_tiles - range + RANGES - 1) / RANGES) : 0;   // tiles range, range + RANGES, ...
    if (n_my == 0) return;
    auto tile_of = [&](int k) { return static_cast<int64_t>(range) + static_cast<int64_t>(k) * RANGES; };

    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- service waves: thread -> hyperedge row, member columns 4 o .. and 64 + 4 o ..; epilogue: output columns 4 o .. of the half
        const int st = tid - 256, row = st >> 4, o = st & 15;
        const int64_t last_pos = n_edges * 3 - 1;
        const uint32_t ldh = static_cast<uint32_t>(ld_h), ldp = static_cast<uint32_t>(ld_p);
        auto fetch_id = [&](int k) {                                     // (st < 48; the tile exists) - scalar tile base, one vector min
            const int64_t first = tile_of(k) * (3 * TE);
            const int lim = static_cast<int>(std::min<int64_t>(last_pos - first, 3 * TE - 1));
            return (i3 + first)[std::min(st, lim)];
        };
        auto load_members = [&](int k, v4f (&hm)[ZX][3]) {
            const int* idk = ids[k & 7] + row * 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float* hp = row_at(h, idk[m], ldh) + 4 * o;
#pragma unroll
                for (int x = 0; x < ZX; ++x) hm[x][m] = *reinterpret_cast<const v4f*>(hp + 64 * x);
            }
        };
        auto load_first_order = [&](int k, v4f (&pr)[3]) {
            const int* idk = ids[k & 7] + row * 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) pr[m] = *reinterpret_cast<const v4f*>(row_at(p, idk[m], ldp) + HC * half + 4 * o);
        };
        auto split_tile = [&](const v4f (&hm)[ZX][3], int buf) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int x = 0; x < ZX; ++x) {
                const v4f u = hm[x][0], q = hm[x][1], it = hm[x][2];
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const v4f z = b == 0 ? u * q : b == 1 ? q * it : b == 2 ? it * u : (u * q) * it;
                    unsigned w0[3], w1[3];
#pragma unroll
                    for (int hp2 = 0; hp2 < 2; ++hp2) split_pair(z[2 * hp2], z[2 * hp2 + 1], hp2 == 0 ? w0 : w1);
                    // columns b D + 64 x + 4 o ..: chunk b D / 8 + 8 x + (o >> 1), half o & 1
                    const int off = row * ZRB + ((((D / 8) * b + 8 * x + (o >> 1)) ^ row) << 4) + 8 * (o & 1);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + pl * ZPL + off) = v2u{w0[pl], w1[pl]};
                }
            }
        };
        auto epilogue = [&](int k, const v4f (&pr)[3]) {                 // tile k - 1
            const float (*pp)[TE][PS] = part[(k - 1) & 1];
            const v4f s0 = *reinterpret_cast<const v4f*>(&pp[0][row][4 * o]), s1 = *reinterpret_cast<const v4f*>(&pp[1][row][4 * o]);
            const v4f s2 = *reinterpret_cast<const v4f*>(&pp[2][row][4 * o]);
            v4f sum = (s0 + s1) + s2;
            if (NBLK == 4) sum = (s0 + s1) + (s2 + *reinterpret_cast<const v4f*>(&pp[3][row][4 * o]));
            const v4f first = (pr[0] + pr[1]) + pr[2];
            const int64_t e = tile_of(k - 1) * TE + row;
            if (e < n_edges) store_stream4(out + e * ld_out + HC * half + 4 * o, sum + first);
        };
        if (st < 3 * TE) {
            ids[0][st] = fetch_id(0);
#pragma unroll
            for (int k = 1; k < 8; ++k) ids[k][st] = k < 4 && k < n_my ? fetch_id(k) : 0;       // (slots of tiles past the end: row 0, requested and dropped)
        }
        __syncthreads();
        v4f hm0[ZX][3], hm1[ZX][3], pr0[3], pr1[3];                        // member values of tile m in hm<m & 1>, first-order rows in pr<m & 1>
        load_members(0, hm0);
        if (n_my > 1) load_members(1, hm1);
        split_tile(hm0, 0);
        __syncthreads();
        int id_carry = 0;
        // phase k: products of tile k + 1 (`use`) into the images; sums, first-order rows and store of tile k - 1; requests: member values of
        // tile k + 2 (`fill`), first-order rows of tile k, ids of tile k + 4 (they reach the ring in the next phase and are first read in the one after)
        auto phase = [&](int k, v4f (&use)[ZX][3], v4f (&fill)[ZX][3], v4f (&pr_cur)[3], v4f (&pr_prev)[3]) {
            if (k >= 1 && k + 3 < n_my && st < 3 * TE) ids[(k + 3) & 7][st] = id_carry;
            load_members(k + 2, fill);                                   // unconditional (past the end: whatever rows the ring slot names, dropped)
            load_first_order(k, pr_cur);
            if (k + 4 < n_my && st < 3 * TE) id_carry = fetch_id(k + 4);
            if (k + 1 < n_my) split_tile(use, (k + 1) & 1);
            // delivery of this phase's requests, THEN the store: the memory counter is in order, a wait behind the store would sit out its
            // round trip to memory in every phase (measured: 2.07 ms instead of 0.9 for this kernel)
            asm volatile("" : "+v"(fill[ZX - 1][0]), "+v"(fill[ZX - 1][1]), "+v"(fill[ZX - 1][2]));      // (in order: the last delivered = all delivered)
            asm volatile("" : "+v"(pr_cur[0]), "+v"(pr_cur[1]), "+v"(pr_cur[2]));
            if (k >= 1) epilogue(k, pr_prev);
            __syncthreads();
        };
        int k = 0;
#pragma clang loop unroll(disable)
        for (; k + 1 <= n_my; k += 2) {                                   // exactly two phases per trip: the register sets come back in place
            phase(k, hm1, hm0, pr0, pr1);
            phase(k + 1, hm0, hm1, pr1, pr0);
        }
        if (k <= n_my) phase(k, hm1, hm0, pr0, pr1);
        return;
    }

    // ---------------- matrix waves: wave = product block; weight planes of the half's 64 output columns x the block's 128 contraction values
    const int blk = wave;
    v8s wreg[4][KB][3];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                wreg[jt][kb][pl] = __builtin_bit_cast(v8s, wsp[(static_cast<int64_t>(((half * 4 + blk) * 4 + jt) * KB + kb) * 3 + pl) * kWave + lane]);
    __syncthreads();
    __syncthreads();
    const int arow = lane & 15, kq = lane >> 4;
    for (int k = 0; k <= n_my; ++k) {
        if (k < n_my && blk < NBLK) {
            v4f acc[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) acc[jt] = v4f{0.f, 0.f, 0.f, 0.f};
            const unsigned char* zp = &zplanes[k & 1][0][0][0] + arow * ZRB;
            v8s zf[KB][3];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) zf[kb][pl] = *reinterpret_cast<const v8s*>(zp + pl * ZPL + ((((D / 8) * blk + 4 * kb + kq) ^ arow) << 4));
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
                        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[jt][kb][kTermB[term]], zf[kb][kTermA[term]], acc[jt], 0, 0, 0);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) *reinterpret_cast<v4f*>(&part[k & 1][blk][arow][16 * jt + 4 * kq]) = acc[jt];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Forward at d = 128 and d = 256 in PASSES over the contraction index (the hyperedge form: `rows=` subsets, IHG_NODE_LEVEL_FORWARD=0).
// The column-half form (kept for d = 64, where one workgroup holds the whole weight block) forms and splits every product twice (once per half); here a workgroup owns ALL 128 output columns and HALF of
// the contraction index - pass A: blocks uq, qi (+ the first-order rows), pass B: blocks iu, uqi, added onto pass A's result, which
// goes through `out` (one extra write and read of [E, d]: 2.2 GB that an issue-bound kernel moves beside its MFMAs) - so every product
// is formed and split ONCE.  The weight planes of a pass are again 192 KB: matrix wave m owns output columns 32 m .. 32 m + 31 with the
// pass's whole contraction index (two blocks = 256 values: 192 registers), so no partial sums meet across waves; it reads all of the
// tile's product images (32 hyperedges x 512 B x 3 planes).  Tiles of 32 hyperedges: half the barriers, ids and address arithmetic per
// hyperedge of the 16-hyperedge form.  Service thread = (hyperedge row, 16 columns): 12 member loads, 32 products, 24 8-byte image
// writes per tile; the epilogue (tile k - 1: sums from the LDS image + first-order rows or pass A's row -> store) rides in phase k.
// A workgroup takes a CONTIGUOUS tile range (hyperedges are numbered by user).
// wkp[pass][m][jt < 2][kb < 8][plane][lane][8]: element i = plane of W[32 m + 16 jt + (lane & 15)][(3 + 2 pass + (kb >> 2)) d + 32 (kb & 3) + 8 (lane >> 4) + i]
// ------------------------------------------------------------------------------------------------
// d = 256 (same kernel, template D): a workgroup owns a column HALF (128 output columns, so both halves of a tile range - adjacent
// workgroups on one XCD - form the pass's products) and a pass is ONE product block (256 values of the contraction index: the same
// 192 weight registers per matrix wave, the same 512-byte image rows); tiles of 16 hyperedges (a service thread = one hyperedge row x 16
// of its 256 member columns); four passes at order 3.  Unlike the chunked kernel below nothing streams the weight planes from L2
// (there: 768 KB per 32-hyperedge tile and workgroup, as much time on the CU's vector-memory path as the tile's MFMAs take).
// wkp, d = 256: [pass = block][half][m][jt < 2][kb < 8][plane][lane][8]: element i = plane of W[128 half + 32 m + 16 jt + (lane & 15)][(3 + pass) d + 32 kb + 8 (lane >> 4) + i]
constexpr int kKpPassV4 = 4 * 2 * 8 * 3 * kWave;                        // v4u of one (pass, column half)'s planes

__global__ __launch_bounds__(kBlockThreads) void pack_planes_fwd_kpass_kernel(const float* __restrict__ w, int64_t ld_w, int d, int nblk, v4u* __restrict__ wkp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int halves = d / 128, passes = d == 128 ? 2 : 4;
    if (idx >= passes * halves * 4 * 2 * 8 * kWave) return;
    const int lane = idx & 63, kb = (idx >> 6) & 7, jt = (idx >> 9) & 1, m = (idx >> 10) & 3, half = (idx >> 12) & (halves - 1), pass = idx >> (d == 128 ? 12 : 13);
    const int b = d == 128 ? 2 * pass + (kb >> 2) : pass;
    const int k0 = d == 128 ? 32 * (kb & 3) : 32 * kb;
    const float* src = w + static_cast<int64_t>(128 * half + 32 * m + 16 * jt + (lane & 15)) * ld_w + (3 + b) * d + k0 + 8 * (lane >> 4);
    const Planes pl = b < nblk ? split8(v4f{src[0], src[1], src[2], src[3]}, v4f{src[4], src[5], src[6], src[7]})
                               : split8(v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int p = 0; p < 3; ++p) wkp[(static_cast<int64_t>(idx >> 6) * 3 + p) * kWave + lane] = pl.p[p];
}

// D: feature width (128: all columns in one workgroup, tiles of 32; 256: column halves, tiles of 16);
// NB: product blocks of this pass (d = 128: 2, or 1 for the second pass of order 2; d = 256: 1); B0: the pass's first block;
// ACC: `out` already holds the earlier passes' result (and the first-order rows) - add onto it instead of gathering the first-order rows
template <int D, int NB, int B0, bool ACC>
__global__ __launch_bounds__(kSplitThreads) void interact_fwd_split_kpass_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p,
                                                                                 const int32_t* __restrict__ i3, const v4u* __restrict__ wkp, float* __restrict__ out,
                                                                                 int64_t ld_out, int64_t n_edges) {
    static_assert((D == 128 && (NB == 1 || NB == 2)) || (D == 256 && NB == 1), "shapes");
    constexpr int HALVES = D / 128, TE = D == 128 ? 32 : 16, RT = TE / 16, TPR = 256 / TE, CSTR = 4 * TPR;      // threads per hyperedge row; stride of a thread's 4-column groups
    constexpr int KB = NB * D / 32, ZRB = 2 * NB * D, ZPL = TE * ZRB, PS = 128 + 4, ZX = D / (4 * TPR), OX = 128 / CSTR;
    __shared__ __attribute__((aligned(16))) unsigned char zplanes[2][3][TE][ZRB];
    __shared__ __attribute__((aligned(16))) float part[2][TE][PS];
    __shared__ int ids[8][3 * TE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int half = HALVES == 1 ? 0 : (bid >> 3) & 1;
    const int range = HALVES == 1 ? bid : (bid & 7) + 8 * (bid >> 4), n_ranges = gridDim.x / HALVES;
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t per = (n_tiles + n_ranges - 1) / n_ranges;
    const int64_t t0 = static_cast<int64_t>(range) * per;
    const int n_my = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(per, n_tiles - t0)));
    if (n_my == 0) return;
    const int hoff = 128 * half;                                         // first output column of this workgroup

    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- service waves: thread -> hyperedge row, member columns 4 o + CSTR x .. (x < ZX), output columns hoff + 4 o + CSTR x .. (x < OX)
        const int st = tid - 256, row = st / TPR, o = st % TPR;
        const int64_t last_pos = n_edges * 3 - 1;
        const uint32_t ldh = static_cast<uint32_t>(ld_h), ldp = static_cast<uint32_t>(ld_p);
        auto fetch_id = [&](int k) {                                     // (st < 3 TE; the tile exists) - scalar tile base, one vector min
            const int64_t first = (t0 + k) * (3 * TE);
            const int lim = static_cast<int>(std::min<int64_t>(last_pos - first, 3 * TE - 1));
            return (i3 + first)[std::min(st, lim)];
        };
        auto load_members = [&](int k, v4f (&hm)[ZX][3]) {
            const int* idk = ids[k & 7] + row * 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float* hp = row_at(h, idk[m], ldh) + 4 * o;
#pragma unroll
                for (int x = 0; x < ZX; ++x) {
                    hm[x][m] = *reinterpret_cast<const v4f*>(hp + CSTR * x);
                }
            }
        };
        // what the products are added to: the three first-order rows (summed in the order u, q, i) or the row the earlier passes left in `out`
        auto load_first = [&](int k, v4f (&pr)[OX][ACC ? 1 : 3]) {
            if (ACC) {
                const int64_t e = std::min<int64_t>((t0 + k) * TE + row, n_edges - 1);
                const float* op = out + e * ld_out + hoff + 4 * o;
#pragma unroll
                for (int x = 0; x < OX; ++x) {
                    pr[x][0] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(op + CSTR * x));     // read once, never again (pass B at C3: 861 -> 847 us;
                }                                                                                         //  the user rows read the same way: 878, they are re-read by the next hyperedges)
            } else {
                const int* idk = ids[k & 7] + row * 3;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const float* pp = row_at(p, idk[m], ldp) + hoff + 4 * o;
#pragma unroll
                    for (int x = 0; x < OX; ++x) pr[x][ACC ? 0 : m] = *reinterpret_cast<const v4f*>(pp + CSTR * x);
                }
            }
        };
        auto split_tile = [&](const v4f (&hm)[ZX][3], int buf) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int x = 0; x < ZX; ++x) {
                const v4f u = hm[x][0], q = hm[x][1], it = hm[x][2];
#pragma unroll
                for (int b2 = 0; b2 < NB; ++b2) {
                    const int b = B0 + b2;
                    const v4f z = b == 0 ? u * q : b == 1 ? q * it : b == 2 ? it * u : (u * q) * it;
                    unsigned w0[3], w1[3];
                    split_pair(z[0], z[1], w0);
                    split_pair(z[2], z[3], w1);
                    // columns b2 D + CSTR x + 4 o ..: chunk b2 D / 8 + (CSTR / 8) x + (o >> 1), half o & 1
                    const int off = row * ZRB + ((((D / 8) * b2 + (CSTR / 8) * x + (o >> 1)) ^ (row & 15)) << 4) + 8 * (o & 1);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + pl * ZPL + off) = v2u{w0[pl], w1[pl]};
                }
            }
        };
        auto epilogue = [&](int k, const v4f (&pr)[OX][ACC ? 1 : 3]) {   // tile k
            const int64_t e = (t0 + k) * TE + row;
            const float (*pp)[PS] = part[k & 1];
#pragma unroll
            for (int x = 0; x < OX; ++x) {
                const v4f sum = *reinterpret_cast<const v4f*>(&pp[row][4 * o + CSTR * x]);
                const v4f first = ACC ? pr[x][0] : (pr[x][0] + pr[x][ACC ? 0 : 1]) + pr[x][ACC ? 0 : 2];
                if (e < n_edges) store_stream4(out + e * ld_out + hoff + 4 * o + CSTR * x, first + sum);
            }
        };
        if (st < 3 * TE) {
            ids[0][st] = fetch_id(0);
#pragma unroll
            for (int k = 1; k < 8; ++k) ids[k][st] = k < 4 && k < n_my ? fetch_id(k) : 0;       // (slots of tiles past the end: row 0, requested and dropped)
        }
        __syncthreads();
        v4f hm0[ZX][3], hm1[ZX][3], pr[OX][ACC ? 1 : 3];                    // member values of tile m in hm<m & 1>
        load_members(0, hm0);
        if (n_my > 1) load_members(1, hm1);
        split_tile(hm0, 0);
        __syncthreads();
        int id_carry = 0;
        // phase k: products of tile k + 1 (`use`) into the images; first-order rows (requested at the start of the phase), sums and store of
        // tile k - 1 at its end; requests: member values of tile k + 2 (`fill`), ids of tile k + 4
        auto phase = [&](int k, v4f (&use)[ZX][3], v4f (&fill)[ZX][3]) {
            if (k >= 1 && k + 3 < n_my && st < 3 * TE) ids[(k + 3) & 7][st] = id_carry;
            load_members(k + 2, fill);                                   // unconditional (past the end: whatever rows the ring slot names, dropped)
            if (k >= 1) load_first(k - 1, pr);
            if (k + 4 < n_my && st < 3 * TE) id_carry = fetch_id(k + 4);
            if (k + 1 < n_my) split_tile(use, (k + 1) & 1);
            // delivery of this phase's requests, THEN the store (the memory counter is in order)
            asm volatile("" : "+v"(fill[ZX - 1][0]), "+v"(fill[ZX - 1][1]), "+v"(fill[ZX - 1][2]));
            asm volatile("" : "+v"(pr[OX - 1][0]));
            if (k >= 1) epilogue(k - 1, pr);
            __syncthreads();
        };
        int k = 0;
#pragma clang loop unroll(disable)
        for (; k + 1 <= n_my; k += 2) {                                   // exactly two phases per trip: the register sets come back in place
            phase(k, hm1, hm0);
            phase(k + 1, hm0, hm1);
        }
        if (k <= n_my) phase(k, hm1, hm0);
        return;
    }

    // ---------------- matrix waves: wave m = output columns hoff + 32 m .. + 31, the pass's whole contraction index
    v8s wreg[2][KB][3];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                wreg[jt][kb][pl] = __builtin_bit_cast(v8s, wkp[(static_cast<int64_t>(((half * 4 + wave) * 2 + jt) * 8 + kb) * 3 + pl) * kWave + lane]);
    __syncthreads();
    __syncthreads();
    const int arow = lane & 15, kq = lane >> 4;
    for (int k = 0; k <= n_my; ++k) {
        if (k < n_my) {
            const unsigned char* zp = &zplanes[k & 1][0][0][0];
            // the fragments of step s + 1 are requested before the MFMAs of step s (IHG_PIN_ORDER)
            auto fragment = [&](int step, v8s (&a)[3]) {
                const int rt = step / KB, kb = step % KB;
                const unsigned char* src = zp + (16 * rt + arow) * ZRB + (((4 * kb + kq) ^ arow) << 4);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const v8s*>(src + pl * ZPL);
            };
            v4f acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) acc[rt][jt] = v4f{0.f, 0.f, 0.f, 0.f};
            v8s a[3], an[3];
            fragment(0, a);
#pragma unroll
            for (int step = 0; step < RT * KB; ++step) {
                const int rt = step / KB, kb = step % KB;
                if (step + 1 < RT * KB) fragment(step + 1, an);
                IHG_PIN_ORDER();                                         // (the scheduler would sink the reads to their first use)
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt)
                        acc[rt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[jt][kb][kTermB[term]], a[kTermA[term]], acc[rt][jt], 0, 0, 0);
                IHG_PIN_ORDER();
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = an[pl];
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) *reinterpret_cast<v4f*>(&part[k & 1][16 * rt + arow][32 * wave + 16 * jt + 4 * kq]) = acc[rt][jt];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradients dW_b[j][c] = sum_e dout[e][j] z_b[e][c].  The contraction runs over the hyperedges, both operands are streams:
// per tile of 32 hyperedges (one MFMA k-block) the dout values and the products z_b of a column HALF (the two halves of a tile range
// are two workgroups on one XCD) are split and laid down as bf16 images, ROW-major
// as they come ([hyperedge][column], 16-byte chunk ch of a row at ch ^ (((row & 3) << 2) | ((row >> 2) & 3)) within each 256-byte
// segment); the MFMA operands need 8 consecutive HYPEREDGES of one column per lane, which ds_read_b64_tr_b16 delivers from those
// images (lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. of a 4 x 16 block and receives column `lane` of its 4
// rows; with this swizzle the reads of a half wave - two blocks 8 rows apart - are conflict-free).  The gradient stays in
// accumulators for the whole kernel; nothing leaves the CU until the end (one slab per tile range, summed by interact.hip's
// slab_reduce_kernel in a fixed order).  One barrier per tile, images double-buffered, rows requested two tiles ahead.
// ------------------------------------------------------------------------------------------------

// The two jobs sit on different waves: waves 0-3 (one per SIMD) only read fragments and issue MFMAs - 192 per tile, block b = wave, all
// eight 16-row tiles of dout columns - and waves 4-7 (their SIMD partners) only request, multiply, split and lay down the next tile.
// An in-order wave that does both stalls its MFMA stream on every wait of the staging work (that form measured 1,880 us against
// 1,710 on the same box); apart, the matrix waves run at the one-wave-per-SIMD rate (tools/split_probe.hip: 17.6 instead of 21
// cycles per MFMA) and part of the split waves' vector instructions falls into the issue cycles the MFMAs leave.  Alone, the matrix
// waves would take 1,350 us and the split waves 1,035.
// D = 128: two column halves of 64 per tile range; D = 256: eight parts of 32 columns (the dout tile is split by every part); D = 64: one workgroup.
// NBLK = 3 (order 2): three product blocks, matrix wave 3 only keeps the barriers.
template <int D, int NBLK>
__global__ __launch_bounds__(kSplitThreads) void interact_bwd_weight_split_ws_kernel(const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3,
                                                                                     const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ slabs,
                                                                                     int64_t n_edges) {
    constexpr int TE = kSplitTE, PARTS = D == 64 ? 1 : (D == 128 ? 2 : 8), RANGES = 256 / PARTS, HC = D / PARTS, CT = HC / 16, JT = D / 16;
    constexpr int DRB = 2 * D < 256 ? 256 : 2 * D, ZRB = 8 * HC;         // image rows are whole 256-byte segments (the transposed-read swizzle moves chunks inside one)
    constexpr int DOCT = D / 64, ZX = HC / 32;                          // per service thread: dout octets, 4-column groups of products
    constexpr int DPL = TE * DRB, ZPL = TE * ZRB;
    __shared__ __attribute__((aligned(16))) unsigned char dplanes[2][3][TE][DRB];
    __shared__ __attribute__((aligned(16))) unsigned char zplanes[2][3][TE][ZRB];
    __shared__ int ids[8][3 * TE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int half = (bid >> 3) & (PARTS - 1), range = (bid & 7) + 8 * (bid / (8 * PARTS));      // `half`: this workgroup's column part
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t per = (n_tiles + RANGES - 1) / RANGES;
    const int64_t t0 = range * per;
    const int n_my = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(per, n_tiles - t0)));

    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- split waves: 256 threads, thread -> hyperedge row, dout octets o and o + 8, member columns 4 o .. and 32 + 4 o ..
        const int st = tid - 256, row = st >> 3, o = st & 7;
        const int64_t last_pos = n_edges * 3 - 1;
        const uint32_t ldh = static_cast<uint32_t>(ld_h), ldd = static_cast<uint32_t>(ld_dout);
        auto fetch_id = [&](int k) {                                     // (st < 96; the tile exists) - scalar tile base, one vector min
            const int64_t first = (t0 + k) * (3 * TE);
            const int lim = static_cast<int>(std::min<int64_t>(last_pos - first, 3 * TE - 1));
            return (i3 + first)[std::min(st, lim)];
        };
        struct Rows {
            v4f d[2 * DOCT], m[ZX][3];
        };
        auto load_rows = [&](int k, Rows& r) {
            const int64_t first = std::min<int64_t>((t0 + k) * TE, n_edges - 1);                    // scalar; tiles past the end: the last row
            const int rows = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(n_edges - (t0 + k) * TE, TE)));
            const float* src = row_at(dout + first * ld_dout, std::min(row, std::max(rows - 1, 0)), ldd) + 8 * o;
#pragma unroll
            for (int x = 0; x < DOCT; ++x) {
                r.d[2 * x] = *reinterpret_cast<const v4f*>(src + 64 * x);
                r.d[2 * x + 1] = *reinterpret_cast<const v4f*>(src + 64 * x + 4);
            }
            if (row >= rows) {                                           // hyperedges past the end contribute nothing
#pragma unroll
                for (int x = 0; x < 2 * DOCT; ++x) r.d[x] = v4f{0.f, 0.f, 0.f, 0.f};
            }
            const int* idk = ids[k & 7] + row * 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float* hp = row_at(h, idk[m], ldh) + HC * half + 4 * o;
#pragma unroll
                for (int x = 0; x < ZX; ++x) r.m[x][m] = *reinterpret_cast<const v4f*>(hp + 32 * x);
            }
        };
        const int swz = tr_swizzle(row);
        auto split_tile = [&](const Rows& r, int buf) {
            auto pair = [&](float xa, float xb, unsigned (&out)[3]) { split_pair(xa, xb, out); };
#pragma unroll
            for (int x = 0; x < DOCT; ++x) {                             // dout octets o + 8 x (256-byte segments swizzled separately)
                v4u sp[3];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    unsigned w[3];
                    pair(r.d[2 * x + (pr >> 1)][2 * (pr & 1)], r.d[2 * x + (pr >> 1)][2 * (pr & 1) + 1], w);
#pragma unroll
                    for (int p = 0; p < 3; ++p) sp[p][pr] = w[p];
                }
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<v4u*>(&dplanes[buf][0][0][0] + p * DPL + row * DRB + 256 * ((o + 8 * x) >> 4) + ((((o + 8 * x) & 15) ^ swz) << 4)) = sp[p];
            }
#pragma unroll
            for (int x = 0; x < ZX; ++x) {                               // member column groups 4 (o + 8 x) ..
                const int og = o + 8 * x;
                const v4f u = r.m[x][0], q = r.m[x][1], it = r.m[x][2];
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const v4f z = b == 0 ? u * q : b == 1 ? q * it : b == 2 ? it * u : (u * q) * it;
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    unsigned w0[3], w1[3];
                    pair(z[0], z[1], w0);
                    pair(z[2], z[3], w1);
                    const int byte = 2 * (b * HC + 4 * og);               // columns b HC + 4 og .. of the product image's row
                    const int off = row * ZRB + 256 * (byte >> 8) + ((((byte >> 4) & 15) ^ swz) << 4) + (byte & 8);
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + p * ZPL + off) = v2u{w0[p], w1[p]};
                }
            }
        };
        if (n_my > 0) {
            if (st < 3 * TE) {
                ids[0][st] = fetch_id(0);
#pragma unroll
                for (int k = 1; k < 8; ++k) ids[k][st] = k < 4 && k < n_my ? fetch_id(k) : 0;   // (slots of tiles past the end: row 0, requested and dropped)
            }
            // (only the split waves read the id ring: a barrier among themselves would do; the workgroup barrier keeps the counts equal)
            __syncthreads();
            Rows r0, r1;
            load_rows(0, r0);
            if (n_my > 1) load_rows(1, r1);
            split_tile(r0, 0);
            __syncthreads();
            int id_carry = 0;
            auto phase = [&](int k, Rows& use, Rows& fill) {
                if (k >= 1 && k + 3 < n_my && st < 3 * TE) ids[(k + 3) & 7][st] = id_carry;
                load_rows(k + 2, fill);                                  // unconditional: a branch around requests costs whole-set register copies
                if (k + 4 < n_my && st < 3 * TE) id_carry = fetch_id(k + 4);
                if (k + 1 < n_my) split_tile(use, (k + 1) & 1);
                __syncthreads();
            };
            int k = 0;
#pragma clang loop unroll(disable)
            for (; k + 1 < n_my; k += 2) {                                // exactly two phases per trip: the register sets come back in place
                phase(k, r1, r0);
                phase(k + 1, r0, r1);
            }
            if (k < n_my) phase(k, r1, r0);
        }
        return;
    }

    // ---------------- matrix waves: wave = product block, 8 x 4 accumulator tiles (all 128 dout columns x the block's 64 columns of the half)
    const int blk = wave;
    v4f acc[JT][CT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[jt][ct] = v4f{0.f, 0.f, 0.f, 0.f};
    if (n_my > 0) {
        __syncthreads();
        __syncthreads();
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int rlo = 8 * g + q, rhi = rlo + 4;
        auto a_addr = [&](int r, int jt) { return r * DRB + 256 * (jt >> 3) + (((2 * (jt & 7) + (pp >> 1)) ^ tr_swizzle(r)) << 4) + 8 * (pp & 1); };
        auto b_addr = [&](int r, int ct) {
            const int byte = 2 * (blk * HC + 16 * ct);                   // first column of the tile in the product image's row
            return r * ZRB + 256 * (byte >> 8) + (((((byte >> 4) & 15) + (pp >> 1)) ^ tr_swizzle(r)) << 4) + 8 * (pp & 1);
        };
        for (int k = 0; k < n_my; ++k) {
            const unsigned char* dp = &dplanes[k & 1][0][0][0];
            const unsigned char* zp = &zplanes[k & 1][0][0][0];
            if (blk < NBLK)
#pragma unroll
            for (int jh = 0; jh < JT / 4; ++jh) {
                v8s a[4][3];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[jt][p] = read_tr_fragment(dp + p * DPL + a_addr(rlo, 4 * jh + jt), dp + p * DPL + a_addr(rhi, 4 * jh + jt));
                v8s b[3], bn[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = read_tr_fragment(zp + p * ZPL + b_addr(rlo, 0), zp + p * ZPL + b_addr(rhi, 0));
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (ct + 1 < CT) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) bn[p] = read_tr_fragment(zp + p * ZPL + b_addr(rlo, ct + 1), zp + p * ZPL + b_addr(rhi, ct + 1));
                    }
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int jt = 0; jt < 4; ++jt)
                            acc[4 * jh + jt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[jt][kTermA[term]], b[kTermB[term]], acc[4 * jh + jt][ct], 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = bn[p];
                }
            }
            __syncthreads();
        }
    }
    if (blk >= NBLK) return;
    float* slab = slabs + static_cast<int64_t>(range) * D * NBLK * D;
    const int c = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[static_cast<int64_t>(16 * jt + 4 * kq + r) * NBLK * D + blk * D + HC * half + 16 * ct + c] = acc[jt][ct][r];
}

}  // namespace

// floats of workspace for the weight planes of one direction: laid out for four blocks at either order
int64_t split_plane_floats(int dim, int order) { return (dim == 64 || dim == 128 || dim == 256) && (order == 2 || order == 3) ? (3LL * 4 * dim * dim) / 2 : 0; }


// the translation unit the ablation macros reach (tools/ab_variant.sh): include/ihgnn_hip.h
extern "C" int32_t ihg_ablation_build(void) { return abl::any ? 1 : 0; }
#ifdef IHG_ABL_M_TRACE
extern "C" int ihg_ablation_trace(unsigned long long* out) {             // [2][1024][4] clock stamps of the last member-gradient launch (tools/phase_trace.py)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_trace), sizeof(g_phase_trace), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

bool split_arith_enabled() {                                             // read at every call: tests and the bench switch it in-process
    const char* v = std::getenv("IHG_INTERACT_ARITH");
    return v == nullptr || std::strcmp(v, "f32") != 0;
}

// either form of g at every width (the gathering form: dim 64 and 128, the caller asks ihg_interact_bwd_gathered_supported)
bool split_members_ok(int dim, int order, const float* g, int64_t ld_h, int64_t ld_dout, const float* dout, bool user_reduced) {
    (void)user_reduced;
    return split_arith_enabled() && (dim == 128 || dim == 64 || dim == 256) && (order == 2 || order == 3) && aligned16(g) && aligned16(dout) &&
           ld_ok(ld_h) && ld_ok(ld_dout);
}

namespace {
template <int D, int NBLK>
void launch_members_split_t(const float* h, int64_t ld_h, const int32_t* i3, const v4u* wsp, const float* winv, const float* dout, int64_t ld_dout, float* g, int64_t n_edges,
                            float* dh_user, int64_t ld_dh, float* bnd_val, int32_t* bnd_user, const float* dy_scale, float* dout_store, int64_t ld_store,
                            const float* inv_src, hipStream_t s) {
    if constexpr (D == 256) {
        if (inv_src != nullptr) {                                        // `dout` holds fp16 planes (ihg_edge_gather_sum_planes)
            if (dh_user != nullptr)
                hipLaunchKernelGGL((interact_bwd_members_split_ws_kernel<D, true, NBLK, false, true>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, i3, wsp, winv, dout, ld_dout,
                                   g, n_edges, dh_user, ld_dh, bnd_val, bnd_user, static_cast<const float*>(nullptr), static_cast<float*>(nullptr), int64_t{0}, inv_src);
            else
                hipLaunchKernelGGL((interact_bwd_members_split_ws_kernel<D, false, NBLK, false, true>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, i3, wsp, winv, dout, ld_dout,
                                   g, n_edges, static_cast<float*>(nullptr), int64_t{0}, static_cast<float*>(nullptr), static_cast<int32_t*>(nullptr),
                                   static_cast<const float*>(nullptr), static_cast<float*>(nullptr), int64_t{0}, inv_src);
            return;
        }
    }
    if constexpr (D == 128 || D == 64) {
        if (dh_user != nullptr && dout_store != nullptr) {              // `dout` is the node-level cotangent: gathered, summed, stored
            hipLaunchKernelGGL((interact_bwd_members_split_ws_kernel<D, true, NBLK, true>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, i3, wsp, winv, dout, ld_dout, g,
                               n_edges, dh_user, ld_dh, bnd_val, bnd_user, dy_scale, dout_store, ld_store);
            return;
        }
    }
    if (dh_user != nullptr) {
        hipLaunchKernelGGL((interact_bwd_members_split_ws_kernel<D, true, NBLK>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, i3, wsp, winv, dout, ld_dout, g, n_edges,
                           dh_user, ld_dh, bnd_val, bnd_user);
        return;
    }
    hipLaunchKernelGGL((interact_bwd_members_split_ws_kernel<D, false, NBLK>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, i3, wsp, winv, dout, ld_dout, g, n_edges,
                       static_cast<float*>(nullptr), int64_t{0}, static_cast<float*>(nullptr), static_cast<int32_t*>(nullptr));
}
}  // namespace

void launch_members_split(int dim, int order, const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, void* planes, const float* dout,
                          int64_t ld_dout, float* g, int64_t n_edges, float* dh_user, int64_t ld_dh, float* bnd_val, int32_t* bnd_user,
                          int* n_boundary_entries, hipStream_t s, const float* dy_scale, float* dout_store, int64_t ld_store, const float* inv_src) {
    v4u* wsp = static_cast<v4u*>(planes);
    const int nblk = order == 3 ? 4 : 3;
    const int items = (dim / 32) * 4 * (dim / 32) * 2 * kWave;
    // the planes (2 fp16 per weight: 4 d^2 dwords) are followed by the weight columns' scales and their inverses ([4][d] floats each)
    float* wsc = reinterpret_cast<float*>(wsp) + 4LL * dim * dim;
    float* winv = wsc + 4 * dim;
    hipLaunchKernelGGL(dense_weight_scales_kernel, dim3(grid_for_waves(static_cast<int64_t>(nblk) * dim)), dim3(kBlockThreads), 0, s, w + 3 * dim, ld_w, int64_t{dim}, nblk, dim, 1,
                       wsc, winv);
    hipLaunchKernelGGL(pack_planes_members_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, nblk, wsc, wsp);
#define IHG_MEMBERS(D)                                                                                                                       \
    {                                                                                                                                        \
        if (nblk == 4) launch_members_split_t<D, 4>(h, ld_h, i3, wsp, winv, dout, ld_dout, g, n_edges, dh_user, ld_dh, bnd_val, bnd_user, dy_scale, dout_store, ld_store, inv_src, s); \
        else launch_members_split_t<D, 3>(h, ld_h, i3, wsp, winv, dout, ld_dout, g, n_edges, dh_user, ld_dh, bnd_val, bnd_user, dy_scale, dout_store, ld_store, inv_src, s);           \
    }
    if (dim == 256) IHG_MEMBERS(256) else if (dim == 64) IHG_MEMBERS(64) else IHG_MEMBERS(128)
#undef IHG_MEMBERS
    if (n_boundary_entries != nullptr) *n_boundary_entries = 2 * (dim == 64 ? 256 : (dim == 128 ? kSplitRanges : 32));     // two per tile range (256 workgroups / column parts: 256, 128, 32 ranges)
}

bool split_weight_ok(int dim, int order, int64_t ld_h, int64_t ld_dout, const float* dout) {
    return split_arith_enabled() && (dim == 64 || dim == 128 || dim == 256) && (order == 2 || order == 3) && aligned16(dout) && ld_ok(ld_h) && ld_ok(ld_dout);
}

int launch_weight_split(int dim, int order, const float* h, int64_t ld_h, const int32_t* i3, const float* dout, int64_t ld_dout, float* slabs, int64_t n_edges,
                        hipStream_t s) {
#define IHG_WEIGHT(D)                                                                                                                                   \
    {                                                                                                                                                   \
        if (order == 3) hipLaunchKernelGGL((interact_bwd_weight_split_ws_kernel<D, 4>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges); \
        else hipLaunchKernelGGL((interact_bwd_weight_split_ws_kernel<D, 3>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges);           \
    }
    if (dim == 256) {
        IHG_WEIGHT(256)
        return 32;
    }
    if (dim == 64) {
        IHG_WEIGHT(64)
        return 256;
    }
    IHG_WEIGHT(128)
#undef IHG_WEIGHT
    return kSplitRanges;                                                 // slabs written (every range writes one, empty ranges zeros)
}

// orders 2 and 3 at d = 64 / 128 / 256
bool split_fwd_ok(int dim, int order, const float* p, int64_t ld_p, const float* out, int64_t ld_out, int64_t ld_h) {
    return split_arith_enabled() && ((order == 3 && (dim == 64 || dim == 128 || dim == 256)) || (order == 2 && (dim == 64 || dim == 128 || dim == 256))) && p != nullptr &&
           ld_ok(ld_p) && ld_out % 4 == 0 && aligned16(p) && aligned16(out) && ld_ok(ld_h);
}

void launch_fwd_split(int dim, int order, const float* h, int64_t ld_h, const float* p, int64_t ld_p, const int32_t* i3, const float* w, int64_t ld_w, void* planes,
                      float* out, int64_t ld_out, int64_t n_edges, hipStream_t s) {
    v4u* wsp = static_cast<v4u*>(planes);
    // d = 128 / 256: passes over the contraction index, every product formed once (the column-half kernel formed them twice; a chunked form that streamed the
    // weight planes at d = 256 filled the CU's L2 port: both removed); d = 64: one workgroup holds the whole weight block
    if (dim == 128 || dim == 256) {
        const int passes = dim == 128 ? 2 : 4, halves = dim / 128;
        hipLaunchKernelGGL(pack_planes_fwd_kpass_kernel, dim3((passes * halves * 4 * 2 * 8 * kWave + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w,
                           dim, order == 3 ? 4 : 3, wsp);
        const int64_t chunk = n_edges;
#define IHG_KPASS(D, NB, B0, ACC, PASS) \
    hipLaunchKernelGGL((interact_fwd_split_kpass_kernel<D, NB, B0, ACC>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, p, ld_p, i3 + 3 * e0, wsp + (PASS) * halves * kKpPassV4, \
                       out + e0 * ld_out, ld_out, std::min<int64_t>(chunk, n_edges - e0))
        for (int64_t e0 = 0; e0 < n_edges; e0 += chunk) {
            if (dim == 128) {
                IHG_KPASS(128, 2, 0, false, 0);
                if (order == 3) IHG_KPASS(128, 2, 2, true, 1);
                else IHG_KPASS(128, 1, 2, true, 1);
            } else {
                IHG_KPASS(256, 1, 0, false, 0);
                IHG_KPASS(256, 1, 1, true, 1);
                IHG_KPASS(256, 1, 2, true, 2);
                IHG_KPASS(256, 1, 3, true, 3);
            }
        }
#undef IHG_KPASS
        return;
    }
    {
        const int items = (dim / 64) * 4 * 4 * (dim / 32) * kWave;
        hipLaunchKernelGGL(pack_planes_fwd_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, order == 3 ? 4 : 3, wsp);
#define IHG_FWD(D)                                                                                                                                              \
    {                                                                                                                                                           \
        if (order == 3) hipLaunchKernelGGL((interact_fwd_split_ws_kernel<D, 4>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, p, ld_p, i3, wsp, out, ld_out, n_edges); \
        else hipLaunchKernelGGL((interact_fwd_split_ws_kernel<D, 3>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, p, ld_p, i3, wsp, out, ld_out, n_edges);           \
    }
        IHG_FWD(64)
#undef IHG_FWD
    }
}

