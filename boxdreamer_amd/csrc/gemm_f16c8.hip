// The GEMM of the F16C8 operand class (include/boxdreamer_hip.h: BD_PREC_F16C8): one f16 MFMA pass + one block-scaled e4m3 correction
// pass per product, in the persistent producer / consumer structure of gemm.hip's gemm_kernel_pc.  Epilogues: gemm_common.h.
#include "gemm_common.h"

#ifdef BD_GEMM_PROBE
extern "C" int bd_gemm_f16c8_probe_set(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bd_probe_buf), &buf, sizeof(buf));
}
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// BD_PREC_F16C8 GEMM (bd_common.h: f16 hi pass + one e4m3 correction pass; 3 bytes per element through LDS-DMA).
// Same producer / consumer persistent structure and 256 x 192 tile as gemm_kernel_pc, with
//   * a stage of four sub-planes per 32-deep slab: A hi (256 x 64 B), W hi (192 x 64 B), A lo8 (256 x 32 B), W lo8 (192 x 32 B)
//     = 42 KiB, so THREE stages fit next to nothing else (the LDS-staged epilogue's 48 KiB scratch overlays stage 2, which is
//     where every tile's last slab lives when K / 32 is a multiple of 3 -- K = 768 and 3072 -- and is free while the next
//     tile's first two slabs land in stages 0 and 1).  With three stages the producers run TWO slabs ahead and wait with a
//     counted vmcnt: the ~1100-cycle issue-to-landing latency of a slab, which bounds the 2-stage kernels at ~2800 cycles per
//     slab in this operand class, is off the critical path.  NSTAGE = 2 (scratch separate) serves the other K.
//   * the e4m3 image q8 of an f16 fragment derived in registers (v_cvt_scalef32_pk_fp8_f16, 4 per fragment), so the
//     correction pass costs one extra ds_read_b128 per 32-row fragment pair instead of two.
//   * (round 5) a SMALL form of the same kernel -- 128 x 192 tiles, 4 consumer + 4 producer waves: ONE consumer wave per SIMD instead of two --
//     for launches whose 256 x 192 tiles would fill their rounds badly (one pose at a time: M = 1536 is 24-96 tiles on 256 CUs): twice the
//     workgroups, each done in half the time.  Same wave tile, same K order, same epilogue arithmetic: a row's result does not depend on the
//     form (tests/test_gpu_ops.py::test_gemm_f16c8_small_form_rows_equal_the_large_form).  (A 128 x 96 form on 2 + 2 waves, two workgroups
//     per CU -- the template takes it: <.., 2, 1, 2> -- is 17 % faster still on fc2 below 3072 rows and slower everywhere else:
//     profiles/r5_f16c8_small_form.md.)
//   * (round 6) the LayerNorm fold (include/boxdreamer_hip.h, bd_gemm_args.ln_*; epilogues in gemm_common.h).  EP 4: the fp32-residual
//     epilogue also emits the rows' F16C8 operand copy and their per-wave-tile (mean, M2) pairs.  LNF: this launch's A operand is such
//     a raw copy; one producer-wave lane per tile row loads the row's eight pairs in front of the tile's first slab (two slabs ahead of the
//     consumers), combines them at once and leaves (rstd, -mean rstd) in an LDS side buffer (double-buffered by tile parity) that the
//     consumers' epilogue reads; the column sums travel like the bias.
//   * (round 6) split-K for launches of a few tiles (one pose at a time: fc2 / proj on 1536 rows are 96 tiles of 128 x 96 on 256 CUs), SK:
//     every tile is computed by S = p.sk_split workgroups (one tile per workgroup, all S on one XCD), workgroup ks taking the slabs
//     [ks nk / S, (ks + 1) nk / S).  The first S - 1 ("secondaries") start from zero and leave their raw accumulators in p.sk_ws, release a
//     per-(tile, ks, wave) flag and exit; the LAST one (highest blockIdx of the tile: dispatched after its secondaries, so its wait cannot
//     starve them) starts from the residual as usual, waits for each flag in turn, adds the partials in the fixed order ks = 0 .. S - 2
//     and runs the epilogue.  Deterministic (fixed association), NOT bit-identical to the unsplit forms: include/boxdreamer_hip.h, sk_ws.
template <int NSTAGE, int EP, int OUTK, bool GELU, int WM = 4, int WN = 2, int NPW = 4, bool LNF = false, bool SK = false>
__global__ __launch_bounds__((WM * WN + NPW) * 64, (WM * WN + NPW) <= 4 ? 2 : 1) void gemm_kernel_pc_f16c8(const bd_gemm_args p) {
    constexpr int MI = 2, NI = 3, NCW = WM * WN;
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32, BK = 32;
    constexpr int A0 = TBM * 64, W0 = TBN * 64, A1 = TBM * 32, W1 = TBN * 64;     // W's e4m3 plane carries q8 AND lo8 (weights: packed once)
    constexpr int OFF_W0 = A0, OFF_A1 = A0 + W0, OFF_W1 = A0 + W0 + A1, STAGE = A0 + W0 + A1 + W1;
    constexpr int SR = 16, SCRATCH = NCW * SR * NI * 32 * 4, S2 = SCRATCH > STAGE ? SCRATCH : STAGE;
    static_assert(SCRATCH <= STAGE, "the epilogue scratch overlays stage 2");
    // producers: every plane of a stage is fetched in 1-KiB pieces (one wave instruction: 16 rows of 64 B, or 32 rows of 32 B), dealt round-robin
    constexpr int GA0 = TBM / 16, GW0 = TBN / 16, GA1 = TBM / 32, GW1 = TBN / 16;
    constexpr int IA0 = (GA0 + NPW - 1) / NPW, IW0 = (GW0 + NPW - 1) / NPW, IA1 = (GA1 + NPW - 1) / NPW, IW1 = (GW1 + NPW - 1) / NPW;
    static_assert(GA0 % NPW == 0 && GW0 % NPW == 0 && GA1 % NPW == 0 && GW1 % NPW == 0, "every producer wave issues the same number of pieces (counted vmcnt)");
    constexpr int PIECES = IA0 + IW0 + IA1 + IW1;      // per slab and producer wave: 12 (256 x 192 on 4 producers), 9 (128 x 192)
    // s_waitcnt vmcnt(PIECES) with expcnt / lgkmcnt left alone (gfx9 encoding: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8])
    constexpr int WAIT_ONE_SLAB = (PIECES & 15) | ((PIECES >> 4) << 14) | 0x70 | 0xF00;
    // side buffer behind the ring + scratch (gemm_kernel_pc): per-column vectors of the current / next tile, q / k RMSNorm weights
    constexpr int AUX_COLP = 2 * STAGE + S2, AUX_RMS = AUX_COLP + 4096, AUX_ROWS = AUX_RMS + 2048, AUX_BYTES = EP == 0 ? 0 : (LNF ? 10240 : 6144);
    static_assert(!LNF || (EP != 0 && NPW * 64 >= TBM && NSTAGE == 3), "LayerNorm fold: one producer lane per tile row");
    static_assert(!SK || ((EP == 3 || EP == 4 || EP == 5) && !LNF && NSTAGE == 3), "split-K: the fp32-residual epilogues, one tile per workgroup");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE + S2 + AUX_BYTES];   // stages at 0, STAGE, 2*STAGE; scratch at 2*STAGE

    bd_saturating_conversions();      // q8 images (K loop) and F16C8 / f16 results (epilogue) saturate instead of turning NaN / inf
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N;
    // split-K: S workgroups per tile, consecutive on one XCD (real id = xcd + 8 (S q + ks) for the virtual workgroup xcd + 8 q)
    const int S = SK ? p.sk_split : 1;
    const int ks = SK ? ((int)blockIdx.x >> 3) % S : 0;
    const int nk_all = p.K / BK;
    const int k_lo = SK ? ks * nk_all / S : 0;
    const int nk = SK ? (ks + 1) * nk_all / S - k_lo : nk_all;
    const int tilesM = (M + TBM - 1) / TBM, tilesN = (N + TBN - 1) / TBN, nt = tilesM * tilesN;
    const int nwg = SK ? (int)gridDim.x / S : (int)gridDim.x;
    const int bid = SK ? ((int)blockIdx.x & 7) + 8 * (((int)blockIdx.x >> 3) / S) : (int)blockIdx.x, xcd = bid & 7;
    const int tq = nt >> 3, tr = nt & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    const int stride = (nwg - xcd + 7) >> 3;
    constexpr int GROUP_M = 4;
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int per_group = GROUP_M * tilesN;
        const int g = t / per_group, in_g = t % per_group;
        const int gm0 = g * GROUP_M;
        const int gh = (tilesM - gm0) < GROUP_M ? (tilesM - gm0) : GROUP_M;
        m0 = (gm0 + in_g % gh) * TBM;
        n0 = (in_g / gh) * TBN;
    };
    const unsigned lds_off = lds_offset_of(lds);

    if (wid >= NCW) {
        // ------------------------------------------------------------------ producers
        const int pw = wid - NCW;
        const unsigned lda0 = (unsigned)(p.lda * 2), ldw0 = (unsigned)(p.ldw * 2), lda1 = (unsigned)p.lda, ldw1 = (unsigned)(p.ldw * 2);
        unsigned oA0[IA0], oW0[IW0], oA1[IA1], oW1[IW1];
#pragma unroll
        for (int i = 0; i < IA0; ++i) { const int row = (pw + NPW * i) * 16 + lane / 4; oA0[i] = row * lda0 + swz_chunk<4>(row, lane % 4) * 16; }
#pragma unroll
        for (int i = 0; i < IW0; ++i) { const int row = (pw + NPW * i) * 16 + lane / 4; oW0[i] = row * ldw0 + swz_chunk<4>(row, lane % 4) * 16; }
#pragma unroll
        for (int i = 0; i < IA1; ++i) { const int row = (pw + NPW * i) * 32 + lane / 2; oA1[i] = row * lda1 + swz_chunk<2>(row, lane % 2) * 16; }
#pragma unroll
        for (int i = 0; i < IW1; ++i) { const int row = (pw + NPW * i) * 16 + lane / 4; oW1[i] = row * ldw1 + swz_chunk<4>(row, lane % 4) * 16; }
        const unsigned char* pA0 = (const unsigned char*)p.A;
        const unsigned char* pW0 = (const unsigned char*)p.W;
        const unsigned char* pA1 = pA0 + p.a_plane * 2;
        const unsigned char* pW1 = pW0 + p.w_plane * 2;
        auto issue = [&](int stage, int m0, int n0, int kt) {
            const int rows_a = (M - m0) < TBM ? (M - m0) : TBM, rows_w = (N - n0) < TBN ? (N - n0) : TBN;
            const unsigned la0 = (unsigned)(rows_a - 1) * lda0 + 48, lw0 = (unsigned)(rows_w - 1) * ldw0 + 48;
            const unsigned la1 = (unsigned)(rows_a - 1) * lda1 + 16, lw1 = (unsigned)(rows_w - 1) * ldw1 + 48;
            const unsigned st = lds_off + stage * STAGE + pw * 1024;
            const unsigned char* ba0 = pA0 + (int64_t)m0 * lda0 + (int64_t)(k_lo + kt) * 64;
            const unsigned char* bw0 = pW0 + (int64_t)n0 * ldw0 + (int64_t)(k_lo + kt) * 64;
            const unsigned char* ba1 = pA1 + (int64_t)m0 * lda1 + (int64_t)(k_lo + kt) * 32;
            const unsigned char* bw1 = pW1 + (int64_t)n0 * ldw1 + (int64_t)(k_lo + kt) * 64;
#pragma unroll
            for (int i = 0; i < IA0; ++i) glds16_s(oA0[i] < la0 ? oA0[i] : la0, ba0, st + i * (NPW * 1024));
#pragma unroll
            for (int i = 0; i < IW0; ++i) glds16_s(oW0[i] < lw0 ? oW0[i] : lw0, bw0, st + OFF_W0 + i * (NPW * 1024));
#pragma unroll
            for (int i = 0; i < IA1; ++i) glds16_s(oA1[i] < la1 ? oA1[i] : la1, ba1, st + OFF_A1 + i * (NPW * 1024));
#pragma unroll
            for (int i = 0; i < IW1; ++i) glds16_s(oW1[i] < lw1 ? oW1[i] : lw1, bw1, st + OFF_W1 + i * (NPW * 1024));
        };
        // issue cursor over this workgroup's slab sequence (all tiles, slab by slab); slab number ig goes to stage ig % NSTAGE
        int ig = 0, ist = 0, it = t_begin + (bid >> 3), ikt = 0, im0 = 0, in0 = 0, itn = 0;
        if (it < t_end) tile_origin(it, im0, in0);
        if constexpr (EP == 2) {
            if (pw == 1) {        // q / k RMSNorm weights (96 floats each), once
                const unsigned off = (unsigned)lane * 16 < 368u ? (unsigned)lane * 16 : 368u;
                glds16_s(off, (const unsigned char*)p.rms_wq, lds_off + AUX_RMS);
                glds16_s(off, (const unsigned char*)p.rms_wk, lds_off + AUX_RMS + 1024);
            }
        }
        int lnt = 0;                       // LNF: tiles whose row statistics this wave has combined (parity = their LDS slot)
        auto issue_next = [&]() {
            if (it >= t_end) return;
            if constexpr (EP != 0) {
                // the tile's bias vector rides in FRONT of its first slab (the counted vmcnt below covers everything but the
                // most recent slab's 12 pieces); the host sends K >= 128 here, so a slot is rewritten only after its tile is done
                if (ikt == 0) {
                    if (pw == 0 && p.bias) {
                        const unsigned off = (unsigned)lane * 16 < (unsigned)(TBN * 4 - 16) ? (unsigned)lane * 16 : (unsigned)(TBN * 4 - 16);
                        glds16_s(off, (const unsigned char*)(p.bias + in0), lds_off + AUX_COLP + (itn & 1) * 2048);
                    }
                    if constexpr (LNF) {
                        if (pw == 1 % NPW) {
                            const unsigned off = (unsigned)lane * 16 < (unsigned)(TBN * 4 - 16) ? (unsigned)lane * 16 : (unsigned)(TBN * 4 - 16);
                            glds16_s(off, (const unsigned char*)(p.ln_colsum + in0), lds_off + AUX_COLP + (itn & 1) * 2048 + 1024);
                        }
                        // The tile's row statistics -> (rstd, -mean rstd) in the LDS side buffer, HERE, two slabs ahead of the consumers:
                        // this wave waits for its four loads right away (and, vmcnt being in order, for the slab it issued last), which the
                        // ring's slack absorbs.  Keeping the loaded registers until the tile starts instead made hipcc guard them with an
                        // s_waitcnt vmcnt(2) in front of every re-load -- a drain of the whole DMA queue once per tile (~1600 cycles per tile,
                        // 11 us on fc1: the probe is in profiles/r6_layernorm_fold.md).  Slot parity = tile parity: the consumers read the
                        // other slot until this tile's first barrier.
                        if (pw * 64 + lane < TBM) {
                            int row = im0 + pw * 64 + lane;
                            row = row < M ? row : M - 1;
                            const f32x4* sp = (const f32x4*)(p.ln_stats_in + (int64_t)row * 16);
                            const f32x4 lnst[4] = {sp[0], sp[1], sp[2], sp[3]};
                            *(float2*)(lds + AUX_ROWS + (lnt & 1) * 2048 + (pw * 64 + lane) * 8) = ln_rows_combine(lnst, p.ln_eps);
                        }
                        ++lnt;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    ++itn;
                }
            }
            issue(ist, im0, in0, ikt);
            ++ig;
            ist = ist + 1 == NSTAGE ? 0 : ist + 1;
            if (++ikt == nk) {
                ikt = 0;
                it += stride;
                if (it < t_end) tile_origin(it, im0, in0);
            }
        };
        // wait until at most `slabs` of this wave's most recent slab fetches are still in flight (12 pieces each)
        auto wait_landed = [&](int slabs) {
            if (slabs <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else { __builtin_amdgcn_s_waitcnt(WAIT_ONE_SLAB); asm volatile("" ::: "memory"); }
        };
#ifdef BD_GEMM_PROBE
        unsigned probe_ts = 0;
#endif
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; ++i) issue_next();
        int g = 0;
        for (int t = t_begin + (bid >> 3); t < t_end; t += stride) {
            for (int kt = 0; kt < nk; ++kt) {
                BD_PROBE_IF(g < 20, g * 3 + 2)
                wait_landed(ig - g - 1);                  // slab g has landed (later slabs may still fly)
                BD_PROBE_IF(g < 20, g * 3)
                pc_barrier();                             // B(g): releases the consumers into slab g; slab g-1 is dead
                BD_PROBE_IF(g < 20, g * 3 + 1)
                if (ig <= g + NSTAGE - 1) issue_next();   // refill the stage slab g-1 lived in
                ++g;
            }
            pc_barrier();                                 // X: consumers are done with the tile's last slab (scratch = stage 2 ..)
        }
#ifdef BD_GEMM_PROBE
        if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wid / WN, wn = wid % WN;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int scale_a = 127, scale_w = 127 - (p.w_qexp + BD_F16C8_D);           // E8M0: the cross terms carry 2^(E + D)
    int g_st = 0;
#ifdef BD_GEMM_PROBE
    unsigned probe_ts = 0;
    int g = 0;
#endif
    BD_PROBE(58) BD_PROBE_RT(56)
    int ti = 0;
    f32x16 acc[MI][NI];
    const bool secondary = SK && ks != S - 1;
    if constexpr (EP != 0) {      // first tile; later tiles are initialised inside the previous tile's epilogue
        int m0, n0;
        tile_origin(t_begin + (bid >> 3), m0, n0);
        if (secondary) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        } else {
            acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
        }
    }
    for (int t = t_begin + (bid >> 3); t < t_end; t += stride, ++ti) {
        int m0, n0;
        tile_origin(t, m0, n0);
        if constexpr (EP == 0) acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
        for (int kt = 0; kt < nk; ++kt) {
            BD_PROBE_IF(g < 20, g * 3)
            pc_barrier();                                 // B
            BD_PROBE_IF(g < 20, g * 3 + 1)
#ifdef BD_GEMM_PROBE
            ++g;
#endif
            const unsigned char* base = lds + g_st * STAGE;
            g_st = g_st + 1 == NSTAGE ? 0 : g_st + 1;
            f16x8 ah[2][MI], wh[2][NI];
            u128 al[MI];
            i32x8 w8[NI];
#define LD16(dst, ptr, row, ks) dst = __builtin_bit_cast(f16x8, *(const u128*)((ptr) + (row) * 64 + (swz_chunk<4>((row), (ks) * 2 + lhalf) << 4)));
#define LDLO(dst, ptr, row) dst = *(const u128*)((ptr) + (row) * 32 + (swz_chunk<2>((row), lhalf) << 4));
            // W's e4m3 operand straight from LDS: 32 bytes = [q8 x 16 | lo8 x 16] of this lane half
#define LDW8(dst, ptr, row)                                                                                    \
            {                                                                                                     \
                const u128 lo_ = *(const u128*)((ptr) + (row) * 64 + (swz_chunk<4>((row), 2 * lhalf) << 4));        \
                const u128 hi_ = *(const u128*)((ptr) + (row) * 64 + (swz_chunk<4>((row), 2 * lhalf + 1) << 4));    \
                dst = (i32x8){(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
            }
            // e4m3 image of one f16 A fragment (8 values of a k-step), 2 dwords; scale 1.0.  The wave runs with MODE.FP16_OVFL set
            // (bd_saturating_conversions): the conversion clamps to +-448, so an activation beyond 448 keeps its full value in the f16
            // pass and only its q_A . lo_W correction term is computed from 448 (bd_common.h, RANGE).
            // (pairs built element-wise: __builtin_bit_cast of a vector ELEMENT to a 2 x f16 vector is miscompiled by hipcc 7.2
            // here -- every conversion then reads the first dword)
#define Q8P(f, a, b) ((h2_){f[a], f[b]})
#define Q8H(d0, d1, f)                                                                                        \
            {                                                                                                     \
                typedef _Float16 h2_ __attribute__((ext_vector_type(2)));                                         \
                typedef short s2_ __attribute__((ext_vector_type(2)));                                            \
                s2_ a_ = {0, 0}, b_ = {0, 0};                                                                     \
                a_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a_, Q8P(f, 0, 1), 1.0f, false);                     \
                a_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a_, Q8P(f, 2, 3), 1.0f, true);                      \
                b_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b_, Q8P(f, 4, 5), 1.0f, false);                     \
                b_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b_, Q8P(f, 6, 7), 1.0f, true);                      \
                d0 = __builtin_bit_cast(unsigned, a_);                                                            \
                d1 = __builtin_bit_cast(unsigned, b_);                                                            \
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) LD16(wh[0][j], base + OFF_W0, wn * (NI * 32) + j * 32 + lrow, 0)
#pragma unroll
            for (int i = 0; i < MI; ++i) LD16(ah[0][i], base, wm * (MI * 32) + i * 32 + lrow, 0)
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NMM = MI * NI;
            static_assert(MI == 2 && NI == 3, "the interleave below is written for 2 x 3 MFMA tiles per wave");
            unsigned qd[MI][4];
#pragma unroll
            for (int q = 0; q < NMM; ++q) {                       // f16, k-step 0
                const int i = q % MI, j = q / MI;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], wh[0][j], acc[i][j], 0, 0, 0);
                if (q < MI) LD16(ah[1][q], base, wm * (MI * 32) + q * 32 + lrow, 1)
                else if (q < MI + NI) LD16(wh[1][q - MI], base + OFF_W0, wn * (NI * 32) + (q - MI) * 32 + lrow, 1)
                if (q == 3) Q8H(qd[0][0], qd[0][1], ah[0][0])
                if (q == 4) Q8H(qd[1][0], qd[1][1], ah[0][1])
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NMM; ++q) {                       // f16, k-step 1
                const int i = q % MI, j = q / MI;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], wh[1][j], acc[i][j], 0, 0, 0);
                if (q < MI) LDLO(al[q], base + OFF_A1, wm * (MI * 32) + q * 32 + lrow)
                if (q == 2) LDW8(w8[0], base + OFF_W1, wn * (NI * 32) + lrow)
                if (q == 4) LDW8(w8[1], base + OFF_W1, wn * (NI * 32) + 32 + lrow)
                if (q == 2) Q8H(qd[0][2], qd[0][3], ah[1][0])
                if (q == 3) Q8H(qd[1][2], qd[1][3], ah[1][1])
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NMM; ++q) {                       // e4m3 correction pass: [lo_A | q_A] . [q_W | lo_W]
                const int i = q % MI, j = q / MI;
                const i32x8 a8 = {(int)al[i][0], (int)al[i][1], (int)al[i][2], (int)al[i][3], (int)qd[i][0], (int)qd[i][1], (int)qd[i][2], (int)qd[i][3]};
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8[j], acc[i][j], 0, 0, 0, scale_a, 0, scale_w);
                if (q == 0) LDW8(w8[2], base + OFF_W1, wn * (NI * 32) + 64 + lrow)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef LDW8
#undef LD16
#undef LDLO
#undef Q8H
#undef Q8P
        }
        BD_PROBE_IF(g == nk, 60)
        pc_barrier();                                     // X
        BD_PROBE_IF(g == nk, 61)
        if constexpr (SK) {
            // flags [tile][3][NCW] ints in the first BD_SPLITK_FLAG_BYTES of sk_ws (zero between launches), then the partials
            // [tile][3][NCW][MI NI 4][64 lanes] x 16 B
            int* const flags = (int*)p.sk_ws;
            f32x4* const parts = (f32x4*)((unsigned char*)p.sk_ws + BD_SPLITK_FLAG_BYTES);
            // Coherence without cache-wide fences: partials and flags move with agent-scope (sc1) loads / stores, which are coherent at the
            // device's coherence point on their own; the secondary waits for its stores' acknowledgements (vmcnt) before it raises the flag,
            // the last workgroup issues its partial loads only after it has seen the flag.  (A release / acquire fence pair here is
            // buffer_wbl2 sc1 + buffer_inv sc1: the invalidate drops the XCD's L2 lines under every workgroup still in its K loop --
            // measured: fc2 at 1536 rows 44 us with S = 2 and SLOWER with S = 3, profiles/r6_split_k.md.)
            if (secondary) {
                const int idx = (t * 3 + ks) * NCW + wid;
                f32x4* dst = parts + (size_t)idx * (MI * NI * 4 * 64) + lane;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + ((i * NI + j) * 4 + q) * 64), "v"(v) : "memory");
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(flags + idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            for (int sk = 0; sk < S - 1; ++sk) {
                const int idx = (t * 3 + sk) * NCW + wid;
                // (bounded: ~1 s of polling.  A secondary of this launch is resident or done by construction -- it was dispatched first --
                // so the bound is never reached in a healthy launch; it turns a lost flag into a wrong tile instead of a hung device.)
                for (int spin = 0; spin < (1 << 24) && __hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin)
                    __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(flags + idx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const f32x4* src = parts + (size_t)idx * (MI * NI * 4 * 64) + lane;
                // the wave tile's 24 x 16 bytes per lane in ONE round trip: the loads go to the coherence point (sc1), ~2 us each way
                // (one wait per four loads -- six dependent round trips -- cost ~9 us per secondary)
                f32x4 v[MI * NI * 4];
#pragma unroll
                for (int u = 0; u < MI * NI * 4; ++u)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(src + u * 64) : "memory");
                static_assert(MI * NI == 6, "24 register quads named below");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                             "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]),
                             "+v"(v[16]), "+v"(v[17]), "+v"(v[18]), "+v"(v[19]), "+v"(v[20]), "+v"(v[21]), "+v"(v[22]), "+v"(v[23]) : : "memory");
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[(i * NI + j) * 4 + q][e];
            }
        }
        unsigned char* scratch = lds + 2 * STAGE + wid * (SR * NI * 32 * 4);
        if constexpr (EP == 0) {
            gemm_epilogue_lds<f16c8, 2, MI, NI, SR, true>(p, acc, scratch, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
        } else {
            const bool has_next = t + stride < t_end;
            int nm0 = 0, nn0 = 0;
            if (has_next) tile_origin(t + stride, nm0, nn0);
            pc_epilogue<f16c8, 2, EP, OUTK, GELU, 2, LNF>(p, acc, (float*)scratch, (float*)scratch + 8 * 96, (const float*)(lds + AUX_COLP + (ti & 1) * 2048),
                                                 (const float*)(lds + AUX_COLP + ((ti + 1) & 1) * 2048), (const float*)(lds + AUX_RMS), wn * (NI * 32), m0 + wm * (MI * 32), n0 + wn * (NI * 32),
                                                 lane, has_next, nm0 + wm * (MI * 32), nn0 + wn * (NI * 32),
                                                 (const float*)(lds + (LNF ? AUX_ROWS : 0) + (ti & 1) * 2048), wm * (MI * 32));
        }
        BD_PROBE_IF(g == nk, 62)
    }
    BD_PROBE(59) BD_PROBE_RT(57)
#ifdef BD_GEMM_PROBE
    if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
}

}  // namespace

int& bd_concurrent_launches() {
    static thread_local int n = 1;
    return n;
}

// Epilogue specialisation of a launch (gemm_kernel_pc's header): 0 generic, 1 / 2 16-bit results (2: fused q/k RMSNorm), 3 fp32 (+ residual)
static int f16c8_epilogue_kind(const bd_gemm_args& a, int& outk, bool& gelu) {
    int ep = 0;
    outk = a.out_f32;
    gelu = a.act == BD_ACT_GELU;
    if (!a.addtab && a.rpg_in <= 0 && !a.wscale && a.N % 192 == 0 && a.K >= 128 && !(a.bias && ((uintptr_t)a.bias & 15))) {
        if (a.out_f32 == OUT_F32) ep = (gelu || a.rms_wq) ? 0 : 3;
        else if (!a.resid && (outk == OUT_OPERAND || outk == OUT_F16 || outk == OUT_BF16X2)) ep = a.rms_wq ? (gelu ? 0 : 2) : 1;
    }
    if (ep == 1 && gelu && outk != OUT_OPERAND) ep = 0;
    return ep;
}

// LayerNorm fold (bd_gemm_args.ln_*): which of the requested sides this class's kernel forms serve.  Producer: the fp32-residual epilogue
// (EP 3 -> 4) of a three-stage launch.  Consumer: K = 768 (three stages), the 16-bit epilogues fc1 (+ GELU), DINOv2's QKV (split-bf16
// planes), BETR's v columns / whole QKV (f16 plane; with the fused q/k RMSNorm).
bool bd_f16c8_takes_ln_fold(const bd_gemm_args& a) {
    int outk = 0;
    bool gelu = false;
    const int ep = f16c8_epilogue_kind(a, outk, gelu);
    const bool s3 = (a.K / 32) % 3 == 0;
    if (!s3 || !wide_epilogue_ok(a, 2)) return false;
    if (ln_fold_producer(a)) {
        if (!ln_fold_producer_ok(a)) return false;
        // the fp32-residual epilogue's launch conditions (f16c8_epilogue_kind's ep == 3), also for the form that reads the residual from the
        // operand copy and may write no fp32 rows at all
        const bool base = !a.addtab && a.rpg_in <= 0 && !a.wscale && a.N % 192 == 0 && a.K >= 128 && !(a.bias && ((uintptr_t)a.bias & 15)) &&
                          !gelu && !a.rms_wq;
        if (!base || (!a.ln_resid_in_op && ep != 3)) return false;
    }
    if (ln_fold_consumer(a)) {
        if (!ln_fold_consumer_ok(a)) return false;
        const bool form = (ep == 1 && gelu && outk == OUT_OPERAND) || (ep == 1 && !gelu && (outk == OUT_F16 || outk == OUT_BF16X2)) ||
                          (ep == 2 && outk == OUT_F16);
        if (!form) return false;
    }
    return true;
}

// Split-K factor of a launch (1 = none).  Only with a scratch region from the caller, only the fp32-residual epilogue kinds on the 128 x 96
// form (one tile per workgroup), only where S workgroups per tile still fit the chip's 2 x CUs slots of that form at once.  args.sk_split
// forces a factor (tests, tuning); otherwise the largest of 2 .. 4 that fits, with at least 6 slabs per workgroup.
static int f16c8_split_k(const bd_gemm_args& a, int ep, bool r5, bool lnp, bool s3, int cus) {
    (void)lnp;                             // (a producer launch that is not r5 has ep == 3: bd_f16c8_takes_ln_fold)
    if (!a.sk_ws || ((uintptr_t)a.sk_ws & 255) || !s3 || a.N % 96 != 0 || !bd_gemm_splitk_flag_bytes(a.M, a.N) || !(ep == 3 || r5)) return 1;
    const int t96 = ((a.M + 127) / 128) * (a.N / 96), nk = a.K / 32;
    // one workgroup per CU: two of this form on a CU put both their consumer-wave pairs on the same two SIMDs (measured: fc2 at 1536 rows
    // 34 us with S = 2 = 192 workgroups, 43 us with S = 3 = 288; profiles/r6_split_k.md)
    const int slots = cus / (bd_concurrent_launches() > 0 ? bd_concurrent_launches() : 1);
    int sk = a.sk_split;
    if (sk <= 0) {
        sk = 1;
        for (int c = 4; c >= 2; --c)
            if (t96 * c <= slots && nk / c >= 6) { sk = c; break; }
    }
    if (sk < 2 || sk > 4 || nk / sk < 1) return 1;
    return sk;
}

extern "C" size_t bd_gemm_splitk_flag_bytes(int M, int N) {
    if (M <= 0 || N <= 0 || N % 96 != 0 || M > BD_SPLITK_MAX_ROWS) return 0;
    const size_t t96 = (size_t)((M + 127) / 128) * (size_t)(N / 96);
    return t96 * 3 * 2 * 4 <= BD_SPLITK_FLAG_BYTES ? BD_SPLITK_FLAG_BYTES : 0;      // [tile][3 secondaries][2 consumer waves] ints
}
extern "C" size_t bd_gemm_splitk_workspace_bytes(int M, int N) {
    const size_t fb = bd_gemm_splitk_flag_bytes(M, N);
    if (!fb) return 0;
    const size_t t96 = (size_t)((M + 127) / 128) * (size_t)(N / 96);
    return fb + t96 * 3 * 2 * (size_t)(64 * 96 * 4);               // + [tile][3][2][64 x 96 fp32 accumulators of a wave tile]
}

// F16C8 has its own persistent kernel; every shape goes through it
int bd_launch_gemm_f16c8(const bd_gemm_args& a, hipStream_t s) {
    if (!wide_epilogue_ok(a, 2) || 256 * a.lda * 2 >= ((int64_t)1 << 31) || 256 * a.ldw * 2 >= ((int64_t)1 << 31)) return BD_ERR_ALIGN;
    if ((a.K % 32) || (a.lda % 32) || (a.ldw % 32)) return BD_ERR_SHAPE;            // the lo8 planes are laid out in 32-element blocks
    if (a.out_f32 == OUT_OPERAND && (a.ldo % 32)) return BD_ERR_SHAPE;
    if (a.w_qexp + BD_F16C8_D < -100 || a.w_qexp + BD_F16C8_D > 120) return BD_ERR_SHAPE;
    if (a.rms_wq && !rms_geometry_ok(a)) return BD_ERR_SHAPE;
    const bool lnp = ln_fold_producer(a), lnc = ln_fold_consumer(a);
    if ((lnp || lnc) && !bd_f16c8_takes_ln_fold(a)) return BD_ERR_SHAPE;
    const int slot = bd_trace_open(s, 0, a.M, a.N, a.K);
    const int cus = cu_count();
    const int tiles = ((a.M + 255) / 256) * ((a.N + 191) / 192);
    // epilogue specialisation (gemm_kernel_pc's header): native / f16 / split-bf16 16-bit results, fp32 (+ residual)
    int outk = 0;
    bool gelu = false;
    const int ep = f16c8_epilogue_kind(a, outk, gelu);
    const bool s3 = (a.K / 32) % 3 == 0;
    // The SMALL form (128 x 192 tiles, one consumer wave per SIMD) where the large tiles would occupy at most half of the CUs -- one pose at
    // a time: M = 1536 is 24-96 large tiles on 256 CUs; with sub-batch lanes, of the CUs' share of one lane.  Measured per shape and row count (profiles/r5_f16c8_small_form.md): up to 128 large
    // tiles the small form is 5-35 % faster, from 144 on it is slower (a lone consumer wave per SIMD needs ~1400 cycles per slab where two
    // need ~2000 for twice the work).  Three-stage ring (K / 32 a multiple of 3: every K of the path) and specialised epilogues only.
    // Every row's result is unchanged.
    const int tiles_small = ((a.M + 127) / 128) * ((a.N + 191) / 192);
    const bool small = s3 && ep != 0 && 2 * tiles * bd_concurrent_launches() <= cus;      // (sub-batch lanes: their launches share the CUs)
    const bool small0 = ep == 0 && 2 * tiles * bd_concurrent_launches() <= cus;             // the generic epilogue's small form (either ring depth)
    const dim3 g0(tiles_small < cus ? tiles_small : cus);
    const int grid = small ? (tiles_small < cus ? tiles_small : cus) : (tiles < cus ? tiles : cus);
    const dim3 g(grid), b(small ? 512 : 768);
#define BD_C8_LAUNCH(EP_, OUTK_, GELU_)                                                                     \
    { if (small) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, EP_, OUTK_, GELU_, 2, 2, 4>), g, b, 0, s, a);     \
      else if (s3) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, EP_, OUTK_, GELU_>), g, b, 0, s, a);            \
      else hipLaunchKernelGGL((gemm_kernel_pc_f16c8<2, EP_, OUTK_, GELU_>), g, b, 0, s, a); }
    // (LayerNorm fold: three-stage forms only -- bd_f16c8_takes_ln_fold)
#define BD_C8_LAUNCH_LN(EP_, OUTK_, GELU_, LNF_)                                                                        \
    { if (small) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, EP_, OUTK_, GELU_, 2, 2, 4, LNF_>), g, b, 0, s, a);         \
      else hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, EP_, OUTK_, GELU_, 4, 2, 4, LNF_>), g, b, 0, s, a); }
    // fc2 (deep K, 4 column tiles) below ~4096 rows: 128 x 96 tiles on 2 + 2 waves -- four times the workgroups of the large form, still at most
    // one per CU -- is another 17 % faster than the 128 x 192 form (52 vs 63 us at 1536 rows; everywhere else it is slower:
    // profiles/r5_f16c8_small_form.md).  One instance (+ its LayerNorm-fold producer twin).
    const bool r5 = lnp && a.ln_resid_in_op;              // the residual comes from (and goes back to) the operand copy: EP 5
    const bool fc2_96 = (ep == 3 || r5) && small && a.K >= 2048 && a.N % 96 == 0 && 4 * tiles * bd_concurrent_launches() <= cus;
    // Split-K (the caller lent a scratch region: bd_gemm_args.sk_ws): the fp32-residual Linears of a launch whose 128 x 96 tiles would still
    // leave most of the chip idle -- one pose at a time, fc2 / proj on 1536 rows: 96 tiles on 256 CUs (two workgroups of this form fit a CU);
    // the last decoder block's query rows: 16 tiles.  S workgroups per tile, see the kernel's header.
    const int sk = f16c8_split_k(a, ep, r5, lnp, s3, cus);
    if (sk > 1) {
        bd_gemm_args b2 = a;
        b2.sk_split = sk;
        const int t96 = ((a.M + 127) / 128) * (a.N / 96);
        const dim3 gs(((t96 + 7) / 8) * 8 * sk), bs(256);
        if (r5 && a.out_f32 == OUT_F32) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 5, OUT_F32, false, 2, 1, 2, false, true>), gs, bs, 0, s, b2);
        else if (r5) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 5, OUT_OPERAND, false, 2, 1, 2, false, true>), gs, bs, 0, s, b2);
        else if (lnp) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 4, OUT_F32, false, 2, 1, 2, false, true>), gs, bs, 0, s, b2);
        else hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 3, OUT_F32, false, 2, 1, 2, false, true>), gs, bs, 0, s, b2);
        bd_trace_close(s, slot);
        BD_CHECK_LAUNCH();
        return BD_OK;
    }
    if (r5 && fc2_96 && a.out_f32 == OUT_F32)
        hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 5, OUT_F32, false, 2, 1, 2>), dim3(((a.M + 127) / 128) * (a.N / 96)), dim3(256), 0, s, a);
    else if (r5 && fc2_96)
        hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 5, OUT_OPERAND, false, 2, 1, 2>), dim3(((a.M + 127) / 128) * (a.N / 96)), dim3(256), 0, s, a);
    else if (r5 && a.out_f32 == OUT_F32) BD_C8_LAUNCH_LN(5, OUT_F32, false, false)
    else if (r5) BD_C8_LAUNCH_LN(5, OUT_OPERAND, false, false)
    else if (lnp && fc2_96)
        hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 4, OUT_F32, false, 2, 1, 2>), dim3(((a.M + 127) / 128) * (a.N / 96)), dim3(256), 0, s, a);
    else if (lnp) BD_C8_LAUNCH_LN(4, OUT_F32, false, false)
    else if (lnc && ep == 1 && gelu) BD_C8_LAUNCH_LN(1, OUT_OPERAND, true, true)
    else if (lnc && ep == 1 && outk == OUT_F16) BD_C8_LAUNCH_LN(1, OUT_F16, false, true)
    else if (lnc && ep == 1) BD_C8_LAUNCH_LN(1, OUT_BF16X2, false, true)
    else if (lnc) BD_C8_LAUNCH_LN(2, OUT_F16, false, true)
    else if (fc2_96)
        hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 3, OUT_F32, false, 2, 1, 2>), dim3(((a.M + 127) / 128) * (a.N / 96)), dim3(256), 0, s, a);
    else if (ep == 3) BD_C8_LAUNCH(3, OUT_F32, false)
    else if (ep == 2 && outk == OUT_OPERAND) BD_C8_LAUNCH(2, OUT_OPERAND, false)
    else if (ep == 2 && outk == OUT_F16) BD_C8_LAUNCH(2, OUT_F16, false)
    else if (ep == 2) BD_C8_LAUNCH(2, OUT_BF16X2, false)
    else if (ep == 1 && gelu) BD_C8_LAUNCH(1, OUT_OPERAND, true)
    else if (ep == 1 && outk == OUT_OPERAND) BD_C8_LAUNCH(1, OUT_OPERAND, false)
    else if (ep == 1 && outk == OUT_F16) BD_C8_LAUNCH(1, OUT_F16, false)
    else if (ep == 1) BD_C8_LAUNCH(1, OUT_BF16X2, false)
    // generic epilogue (table add, row remap, hand-offs between operand classes: the patch / heat-map embeddings, the adapter): round 6 gives
    // it the small form too -- one pose at a time these launches were 24 workgroups of the large form, 64 us each (rows bit-identical)
    else if (small0 && s3) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 0, OUT_OPERAND, false, 2, 2, 4>), g0, dim3(512), 0, s, a);
    else if (small0) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<2, 0, OUT_OPERAND, false, 2, 2, 4>), g0, dim3(512), 0, s, a);
    else if (s3) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, 0, OUT_OPERAND, false>), g, b, 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel_pc_f16c8<2, 0, OUT_OPERAND, false>), g, b, 0, s, a);
#undef BD_C8_LAUNCH
#undef BD_C8_LAUNCH_LN
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}
