// MFMA GEMM with fused epilogue for every nn.Linear on the corner-heatmap path.
//
//   out[map(r), :] = act(wscale * (A[r, :] . W^T) + bias) + addtab[r % tab_rows, :] + resid[map(r), :]
//
// gfx950 design.  Both operands are K-contiguous ("A . W^T", the nn.Linear layout), so A and W tiles share one
// loader.  Tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA: no staging VGPRs, no ds_write pass).  An
// LDS-DMA wave-instruction writes 1 KiB at  M0-base + lane*16  (lane-linear), so the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied on the SOURCE side: the lane that owns LDS slot (row, c')
// fetches logical chunk c = c' ^ swz(row) of that row (same 128-byte line: coalescing unchanged), and fragment reads
// apply the same involution.  One barrier per K-slab (s_waitcnt vmcnt(0) + s_barrier): slab kt landed, and the
// buffer about to be overwritten is free; the DMA for slab kt+1 is issued right after it and flies under slab kt.
// Workgroup ids are remapped (XCD-aware, grouped raster) so that each XCD's L2 sees a compact patch of tiles.
//
// Operand classes (template T):
//   bf16 / f16   v_mfma_f32_32x32x16, slab = 64 k  (128-byte tile rows), a fragment = one 16-byte chunk per lane
//   split-bf16   NS = 2 planes (hi, lo): hi*hi + hi*lo + lo*hi per fragment pair, slab = 32 k
//   e4m3         v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (2x rate), slab = 128 k (128-byte rows),
//                a fragment = two chunks per lane; per-output-channel weight scale applied in the epilogue
// Tile geometry (template): WM x WN waves, each owning an (MI*32) x (NI*32) sub-tile:
//   <2,4,4,2> 256x256, 8 waves, 128 KiB LDS, 1 workgroup/CU (wide outputs: 128 FLOP per LDS-DMA byte)
//   <2,2,2,2> 128x128, 4 waves, 64 KiB, 2 workgroups/CU (N = 768 outputs)      <2,2,1,1> 64x64 (latency mode)
// Experiments that did NOT pay (deeper LDS-DMA rings, mid-slab barriers, ping-pong wave groups, 256x128 tiles at one or
// two workgroups per CU, a persistent tile loop with the epilogue overlapped, pinned fragment prefetch) and the counters
// behind the choices: profiles/r1_gemm_experiments.md, profiles/r1_gemm_pmc.md.
#include "gemm_common.h"

#ifdef BD_GEMM_PROBE
extern "C" int bd_gemm_probe_set(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bd_probe_buf), &buf, sizeof(buf));
}
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// NSTG > 2 (small-batch launches: fewer tiles than the chip has workgroup slots, so a tile's time is the chain of its slabs' DMA
// latencies, not the matrix work): a ring of NSTG stages with NSTG - 1 slabs in flight and counted vmcnt waits; the arithmetic
// (K order per row) is that of the 2-stage form, so a row's result does not depend on which form a launch takes.
template <class T, int NS, int BK, int WM, int WN, int MI, int NI, int NSTG = 2>
__global__ __launch_bounds__(WM * WN * 64, NSTG == 2 ? 2 : 1) void gemm_kernel_glds(const bd_gemm_args p) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 frag_t;
    constexpr int ESZ = OpGeom<T>::ESZ, KSTEP = OpGeom<T>::KSTEP, CPF = OpGeom<T>::CPF;
    constexpr int NWAVE = WM * WN;
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    constexpr int ROWB = BK * ESZ;                    // bytes per tile row per slab
    constexpr int CH = ROWB / 16;                     // 16-byte chunks per tile row
    constexpr int RPP = 64 / CH;                      // tile rows per 1-KiB DMA piece
    constexpr int A_BYTES = TBM * ROWB, W_BYTES = TBN * ROWB;
    constexpr int PPW_A = A_BYTES / 1024 / NWAVE, PPW_W = W_BYTES / 1024 / NWAVE;
    static_assert(PPW_A >= 1 && PPW_W >= 1 && A_BYTES % (1024 * NWAVE) == 0 && W_BYTES % (1024 * NWAVE) == 0, "tile/DMA split");
    constexpr int STAGE_BYTES = (A_BYTES + W_BYTES) * NS;   // A planes then W planes
    constexpr int KS = BK / KSTEP;
    static_assert(KS >= 1 && CH * 16 == ROWB && (CH == 4 || CH == 8), "slab geometry");
    constexpr int EPI_SCRATCH = NWAVE * 32 * NI * 32 * 4;   // the wide epilogue's per-wave transpose scratch
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTG * STAGE_BYTES > EPI_SCRATCH ? NSTG * STAGE_BYTES : EPI_SCRATCH];
    constexpr int PPW = (PPW_A + PPW_W) * NS;         // DMA pieces this wave issues per slab
    static_assert(NSTG >= 2 && NSTG <= 4 && (NSTG - 2) * PPW < 64, "ring depth / vmcnt range");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    int m0, n0;
    tile_coords_t<TBM, TBN>(p.M, p.N, TBM >= 256 ? 4 : 8, m0, n0);

    // per-lane DMA sources (byte pointers): piece j covers tile rows j*RPP .. j*RPP+RPP-1
    const unsigned char* ga[PPW_A];
    const unsigned char* gw[PPW_W];
#pragma unroll
    for (int i = 0; i < PPW_A; ++i) {
        const int row = (wid * PPW_A + i) * RPP + lane / CH;
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
        ga[i] = (const unsigned char*)p.A + ((int64_t)ar * p.lda) * ESZ + swz_chunk<CH>(row, lane % CH) * 16;
    }
#pragma unroll
    for (int i = 0; i < PPW_W; ++i) {
        const int row = (wid * PPW_W + i) * RPP + lane / CH;
        int wr = n0 + row; wr = wr < p.N ? wr : p.N - 1;
        gw[i] = (const unsigned char*)p.W + ((int64_t)wr * p.ldw) * ESZ + swz_chunk<CH>(row, lane % CH) * 16;
    }
    const int64_t a_plane = p.a_plane * ESZ, w_plane = p.w_plane * ESZ;
    const unsigned lds_off = lds_offset_of(lds);
#define DMA_SLAB(buf, kb)                                                                                     \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                          \
        _Pragma("unroll") for (int i = 0; i < PPW_A; ++i)                                                     \
            glds16(ga[i] + s * a_plane + (kb), lds_off + (buf) * STAGE_BYTES + s * A_BYTES + (wid * PPW_A + i) * 1024); \
        _Pragma("unroll") for (int i = 0; i < PPW_W; ++i)                                                     \
            glds16(gw[i] + s * w_plane + (kb), lds_off + (buf) * STAGE_BYTES + NS * A_BYTES + s * W_BYTES + (wid * PPW_W + i) * 1024); \
    }

    f32x16 acc[MI][NI];
    acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);

    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nk = p.K / BK;
#ifdef BD_GEMM_PROBE
    unsigned probe_ts = 0;
#endif
    BD_PROBE(60)
    DMA_SLAB(0, 0)
    if constexpr (NSTG > 2) {
#pragma unroll
        for (int s2 = 1; s2 < NSTG - 1; ++s2)
            if (s2 < nk) { DMA_SLAB(s2, s2 * ROWB) }
    }
    for (int kt = 0; kt < nk; ++kt) {
        BD_PROBE(kt * 3)
        if constexpr (NSTG == 2) {
            slab_barrier();                                // slab kt landed; buffer (kt+1)&1 is free
        } else {
            // everything but the (up to NSTG - 2) younger slabs' pieces of this wave has landed; then everyone else's
            const int younger = nk - 1 - kt < NSTG - 2 ? nk - 1 - kt : NSTG - 2;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        BD_PROBE(kt * 3 + 1)
        if (kt + NSTG - 1 < nk) { DMA_SLAB((kt + NSTG - 1) % NSTG, (kt + NSTG - 1) * ROWB) }      // into the stage of slab kt - 1
        BD_PROBE(kt * 3 + 2)
        const unsigned char* base = lds + (kt % NSTG) * STAGE_BYTES;
        // fragments are double-buffered in registers (one plane only: two sets next to 128 accumulators would
        // spill in the split mode): the LDS reads of k-step ks+1 are in flight while the MFMAs of ks issue
        constexpr int FB = NS == 1 ? 2 : 1;
        frag_t a[FB][NS][MI], b[FB][NS][NI];
        // a fragment = CPF consecutive chunks starting at chunk (ks*2 + lane_half) * CPF of the row
#define LOAD_ONE(dst, ptr, row, ks)                                                                           \
        {                                                                                                     \
            const int c0_ = ((ks) * 2 + lhalf) * CPF;                                                         \
            if constexpr (CPF == 1) {                                                                         \
                dst = __builtin_bit_cast(frag_t, *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4))); \
            } else {                                                                                          \
                const u128 lo_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4));     \
                const u128 hi_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_ + 1) << 4)); \
                dst = (frag_t){(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
            }                                                                                                 \
        }
#define LOAD_FRAGS(ks, slot)                                                                                  \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                      \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                    \
                LOAD_ONE(a[slot][s][i], base + s * A_BYTES, wm * (MI * 32) + i * 32 + lrow, ks)               \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                    \
                LOAD_ONE(b[slot][s][j], base + NS * A_BYTES + s * W_BYTES, wn * (NI * 32) + j * 32 + lrow, ks) \
        }
        if (FB == 2) { LOAD_FRAGS(0, 0) }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = FB == 2 ? (ks & 1) : 0;
            if (FB == 2) { if (ks + 1 < KS) { LOAD_FRAGS(ks + 1, (cur ^ 1) & (FB - 1)) } }
            else { LOAD_FRAGS(ks, 0) }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (NS == 2) {
                        acc[i][j] = Op16<T>::mfma(a[cur][NS - 1][i], b[cur][0][j], acc[i][j]);   // lo * hi
                        acc[i][j] = Op16<T>::mfma(a[cur][0][i], b[cur][NS - 1][j], acc[i][j]);   // hi * lo
                    }
                    acc[i][j] = Op16<T>::mfma(a[cur][0][i], b[cur][0][j], acc[i][j]);            // hi * hi
                }
        }
#undef LOAD_FRAGS
#undef LOAD_ONE
    }
#undef DMA_SLAB
    // wide epilogue needs 16-byte aligned rows: N % 8 == 0 and aligned leading dimensions (else scalar path)
    const bool wide = (p.N % 8 == 0) && (p.ldo % 8 == 0) && (((uintptr_t)p.out & 15) == 0) &&
                      (!p.resid || ((p.ldr % 4 == 0) && ((uintptr_t)p.resid & 15) == 0)) &&
                      (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.addtab || ((uintptr_t)p.addtab & 15) == 0) &&
                      (!p.wscale || ((uintptr_t)p.wscale & 15) == 0) &&
                      (p.out_f32 || NS == 1 || (p.out_plane % 8 == 0));
    BD_PROBE(61)
    if (wide) {
        __syncthreads();                                   // every wave is done with the operand slabs
        BD_PROBE(62)
        gemm_epilogue_lds<T, NS, MI, NI>(p, acc, lds + wid * (32 * NI * 32 * 4), m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    } else {
        gemm_epilogue<T, NS, MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    }
#ifdef BD_GEMM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the epilogue's stores have left
    BD_PROBE(63)
    if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * NWAVE + wid) * 64 + lane] = probe_ts;
#endif
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer, persistent form of the same GEMM  (round 2).
//
// Why (tools/gemm_phase_probe.py, profiles/r2_gemm_phase_probe.md): in gemm_kernel_glds every wave issues its share of the
// next slab's LDS-DMA right after the slab barrier.  The CU's texture-address path accepts one 1-KiB global_load_lds per ~23
// cycles, so the 64 pieces of a 256x256x64 slab keep the ISSUING waves blocked for 580 (older wave of a SIMD) to 1470 cycles
// (younger wave) per slab, during which they issue no MFMA: a slab takes ~3300 cycles for 2048 cycles of matrix work per
// SIMD, and the two waves of a SIMD end up running their MFMA phases one after the other.
//
// Here the workgroup has WM*WN consumer waves (MFMA + epilogue, never a VMEM instruction inside the K loop) and ONE producer
// wave that issues every LDS-DMA piece (SGPR base + 32-bit VGPR offset form) and is the only wave that waits on vmcnt.  One
// s_barrier per slab:  B(kt) = "slab kt has landed (the producer waited for it) and every consumer is done with slab kt-1";
// after B(kt) the producer refills the buffer slab kt-1 lived in.  The kernel is persistent (grid = CUs, static XCD-aware
// tile lists), which buys two more overlaps: the first slab of the NEXT tile is fetched under the last slab of this one, and
// the epilogue's global stores are not waited for by anyone -- they drain under the next tile's MFMAs (in the one-tile kernel
// all CUs write their tiles at the same moment: QKV's 128 KiB per CU sit at the HBM write floor of ~10k cycles while the
// matrix pipes idle).  Barriers per tile: nk (B) + 1 (X: every consumer is done reading the last slab).
//
// Consumers execute NO vector-memory LOAD anywhere in the specialised epilogues (EP 1-3 below): on gfx9 loads and stores share
// the in-order vmcnt, so a load issued after the epilogue's stores (a bias vector, a residual row, a register-spill reload)
// can only be waited for together with those stores' acknowledgements.  The first version of this kernel had exactly that
// (generic epilogue under the 168-VGPR cap: scratch reloads and per-column-block bias loads between the stores) and a LONE
// workgroup spent 13.8k cycles per tile in its epilogue against 1.9k for the bare stores (profiles/r2_gemm_epilogue.md).  Now:
//   * per-column vectors (bias, e4m3 weight scale) of a tile arrive in a small LDS side buffer with the tile's first slab
//     (one extra 1-KiB DMA piece each, producer wave 0; double-buffered by tile parity); the q/k RMSNorm weights once per kernel;
//   * a plain fp32 residual is loaded INTO the accumulators of the NEXT tile (acc_init convention) from inside this tile's
//     epilogue, chunk by chunk as the accumulator registers are flushed to the LDS scratch -- before that chunk's stores;
//   * the epilogue kind is a template parameter (EP), so no path carries the other paths' live values or branches.
// EP: 0 = generic (gemm_epilogue_lds: table add, row remap, residual in scaled modes, odd output kinds),
//     1 = 16-bit / 8-bit result (OUTK: operand-native, f16, bf16), optional GELU,
//     2 = 16-bit result with the fused q/k RMSNorm,
//     3 = fp32 result (+ fp32 residual through the accumulators).
// What still bounds it (profiles/r2_gemm_phase_probe.md): per slab the producers need ~1000 cycles to issue + ~840 to land
// the next slab (a 2-stage ring cannot hide that latency: 2045 cycles per slab for 1536 of matrix work per SIMD).
// Negative results kept in the profile notes: a row-per-lane epilogue on transposed accumulators (no LDS, 2.1x slower: a wave
// store touching 32 rows x 32 bytes is issue-bound in the texture path), non-temporal vs plain stores (equal), start-up
// staggers that spread the CUs' epilogues over the tile period (+1 % / -9 % QKV, -7 % / -2 % fc1).

// RING (round 4): 0 = two symmetric stages (above).  1 = an ASYMMETRIC operand ring: A three slabs deep, W two (3 x 32 + 2 x 24 KiB; three
// whole 56-KiB stages do not fit the 160 KiB).  After B(g) the producers fetch W(g+1) -- 24 pieces, ~550 cycles of issue + the landing
// latency: inside the 1536 cycles of matrix work slab g holds -- and then A(g+2), which has a whole extra slab to land.  The 2-stage
// form needs ~1000 + ~840 cycles for slab g+1 against those 1536, and the counters show the consumers parked at the slab barrier for
// ~28 % of their time (profiles/r4_gemm_counters.md); the F16C8 kernel, whose 3-stage ring of 48-KiB stages fits, shows ~7 %.
// DMA rows stay 128 bytes (round 3's deeper rings shortened them and lost).  The epilogue scratch (8 waves x 6 KiB) is the two operand
// buffers the tile's last slab lived in -- rows 0-7 of a 16-row chunk in the W buffer, rows 8-15 in the A buffer (pc_epilogue: sc, sc_hi).
// LNF (round 6): the consumer side of the LayerNorm fold (include/boxdreamer_hip.h, bd_gemm_args.ln_*; gemm_f16c8.hip's header has the
// mechanism) -- here for the one launch of the default mode this file serves: BETR's q, k columns (f16, fused q/k RMSNorm, RING 1).
template <class T, int NS, int BK, int WM, int WN, int MI, int NI, int NPW, int EP, int OUTK, bool GELU, int RING = 0, bool LNF = false>
__global__ __launch_bounds__((WM * WN + NPW) * 64, 1) void gemm_kernel_pc(const bd_gemm_args p) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 frag_t;
    constexpr int ESZ = OpGeom<T>::ESZ, KSTEP = OpGeom<T>::KSTEP, CPF = OpGeom<T>::CPF;
    constexpr int NCW = WM * WN;                      // consumer waves; waves NCW .. NCW + NPW - 1 are producers
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    constexpr int ROWB = BK * ESZ;
    constexpr int CH = ROWB / 16;
    constexpr int RPP = 64 / CH;
    constexpr int A_BYTES = TBM * ROWB, W_BYTES = TBN * ROWB;
    constexpr int PA = A_BYTES / 1024, PW = W_BYTES / 1024;       // 1-KiB DMA pieces per plane per slab
    constexpr int STAGE_BYTES = (A_BYTES + W_BYTES) * NS;
    constexpr int KS = BK / KSTEP;
    static_assert(KS >= 1 && CH * 16 == ROWB && (CH == 4 || CH == 8) && A_BYTES % 1024 == 0 && W_BYTES % 1024 == 0, "slab geometry");
    static_assert(PA % NPW == 0 && PW % NPW == 0, "pieces split evenly over the producer waves");
    constexpr int SR = (EP == 0 && NCW * 32 * NI * 32 * 4 <= STAGE_BYTES) ? 32 : 16;      // LDS-staged epilogue: scratch rows per pass
    static_assert(NCW * SR * NI * 32 * 4 <= STAGE_BYTES, "the epilogue scratch must fit one stage");
    static_assert(EP == 0 || (MI == 2 && NI == 3 && SR == 16 && TBN <= 256), "pc_epilogue is written for 96-column wave tiles");
    // side buffer behind the ring: per-column vectors of the current / next tile (2 x [bias 1 KiB | weight scale 1 KiB]) and
    // the q / k RMSNorm weights (2 x 1 KiB)
    constexpr int RA = NS * A_BYTES, RW = NS * W_BYTES;            // one slab's A / W image (all planes)
    constexpr int RING_BYTES = RING ? 3 * RA + 2 * RW : 2 * STAGE_BYTES;
    static_assert(RING == 0 || (EP != 0 && NCW * 8 * NI * 32 * 4 <= RW && NCW * 8 * NI * 32 * 4 <= RA), "split scratch: half a chunk per buffer");
    constexpr int AUX_COLP = RING_BYTES, AUX_RMS = AUX_COLP + 4096, AUX_ROWS = AUX_RMS + 2048, AUX_BYTES = EP == 0 ? 0 : (LNF ? 10240 : 6144);
    static_assert(!LNF || (RING == 1 && EP != 0 && NPW * 64 >= TBM && sizeof(T) == 2), "LayerNorm fold: ring form, one producer lane per tile row");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[RING_BYTES + AUX_BYTES];
    // byte offset of slab g's A / W image inside the ring (ga = g % 3, gw = g % 2 in the asymmetric form; both = g & 1 otherwise)
    auto off_a = [](int ga) { return RING ? ga * RA : ga * STAGE_BYTES; };
    auto off_w = [](int gw) { return RING ? 3 * RA + gw * RW : gw * STAGE_BYTES + RA; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N;
    const int nk = p.K / BK;

    // static tile list: XCD x (= blockIdx % 8) owns a contiguous run of logical tile ids (grouped raster, as in
    // tile_coords_t); its workgroups walk that run with stride = workgroups on the XCD, so the CUs of an XCD always work on
    // consecutive tiles (a compact patch of the output: shared A panels / W tiles stay in that XCD's L2).
    const int tilesM = (M + TBM - 1) / TBM, tilesN = (N + TBN - 1) / TBN, nt = tilesM * tilesN;
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
    const int tq = nt >> 3, tr = nt & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    const int stride = (nwg - xcd + 7) >> 3;          // workgroups living on this XCD
    constexpr int GROUP_M = TBM >= 256 ? 4 : 8;
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
        // ------------------------------------------------------------------ producers (wave pw takes pieces pw, pw + NPW, ...)
        const int pw = wid - NCW;
        // Per-lane byte offsets of the 1-KiB pieces inside a tile are the same for every tile (piece j = tile rows
        // j*RPP .. j*RPP+RPP-1, source-side swizzled chunk); the tile origin, the K position and the plane go into the
        // scalar base.  Rows past the edge of the tensor are clamped with one v_min against the tile's last valid byte
        // (they load in-bounds garbage whose results the epilogue discards).
        const unsigned lda_b = (unsigned)(p.lda * ESZ), ldw_b = (unsigned)(p.ldw * ESZ);
        unsigned offA[PA / NPW], offW[PW / NPW];
#pragma unroll
        for (int j = 0; j < PA / NPW; ++j) {
            const int row = (j * NPW + pw) * RPP + lane / CH;
            offA[j] = (unsigned)row * lda_b + swz_chunk<CH>(row, lane % CH) * 16;
        }
#pragma unroll
        for (int j = 0; j < PW / NPW; ++j) {
            const int row = (j * NPW + pw) * RPP + lane / CH;
            offW[j] = (unsigned)row * ldw_b + swz_chunk<CH>(row, lane % CH) * 16;
        }
        const int64_t a_plane = p.a_plane * ESZ, w_plane = p.w_plane * ESZ;
        auto issue_a = [&](int ga, int m0, int kt) {
            const int rows_a = (M - m0) < TBM ? (M - m0) : TBM;
            const unsigned lim_a = (unsigned)(rows_a - 1) * lda_b + (CH - 1) * 16;
            const unsigned st = lds_off + off_a(ga) + pw * 1024;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const unsigned char* ba = (const unsigned char*)p.A + (int64_t)m0 * lda_b + (int64_t)kt * ROWB + s2 * a_plane;
#pragma unroll
                for (int j = 0; j < PA / NPW; ++j)
                    glds16_s(offA[j] < lim_a ? offA[j] : lim_a, ba, st + s2 * A_BYTES + j * NPW * 1024);
            }
        };
        auto issue_w = [&](int gw, int n0, int kt) {
            const int rows_w = (N - n0) < TBN ? (N - n0) : TBN;
            const unsigned lim_w = (unsigned)(rows_w - 1) * ldw_b + (CH - 1) * 16;
            const unsigned st = lds_off + off_w(gw) + pw * 1024;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const unsigned char* bw = (const unsigned char*)p.W + (int64_t)n0 * ldw_b + (int64_t)kt * ROWB + s2 * w_plane;
#pragma unroll
                for (int j = 0; j < PW / NPW; ++j)
                    glds16_s(offW[j] < lim_w ? offW[j] : lim_w, bw, st + s2 * W_BYTES + j * NPW * 1024);
            }
        };
        auto issue = [&](int stage, int m0, int n0, int kt) { issue_a(stage, m0, kt); issue_w(stage, n0, kt); };
        auto issue_colp = [&](int n0, int slot) {     // a tile's per-column vectors (N % TBN == 0: TBN floats are in bounds), producer wave 0
            if (pw == 0) {
                const unsigned off = (unsigned)lane * 16 < (unsigned)(TBN * 4 - 16) ? (unsigned)lane * 16 : (unsigned)(TBN * 4 - 16);
                if (p.bias) glds16_s(off, (const unsigned char*)(p.bias + n0), lds_off + AUX_COLP + (slot & 1) * 2048);
                if (sizeof(T) == 1 && p.wscale) glds16_s(off, (const unsigned char*)(p.wscale + n0), lds_off + AUX_COLP + (slot & 1) * 2048 + 1024);
            }
            if constexpr (LNF) {
                if (pw == 1) {        // the column sums s[n] travel in the (otherwise unused) scale slot
                    const unsigned off = (unsigned)lane * 16 < (unsigned)(TBN * 4 - 16) ? (unsigned)lane * 16 : (unsigned)(TBN * 4 - 16);
                    glds16_s(off, (const unsigned char*)(p.ln_colsum + n0), lds_off + AUX_COLP + (slot & 1) * 2048 + 1024);
                }
            }
        };
        int lnt = 0;                       // LNF: tiles whose row statistics this wave has combined (parity = their LDS slot)
        auto load_row_stats = [&](int m0) {      // (gemm_f16c8.hip has the why: combined at once, not kept in registers until the tile starts)
            if (pw * 64 + lane < TBM) {
                int row = m0 + pw * 64 + lane;
                row = row < M ? row : M - 1;
                const f32x4* sp = (const f32x4*)(p.ln_stats_in + (int64_t)row * 16);
                const f32x4 lnst[4] = {sp[0], sp[1], sp[2], sp[3]};
                *(float2*)(lds + AUX_ROWS + (lnt & 1) * 2048 + (pw * 64 + lane) * 8) = ln_rows_combine(lnst, p.ln_eps);
            }
            ++lnt;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
#ifdef BD_GEMM_PROBE
        unsigned probe_ts = 0;
#endif
        // Issue cursor: slab number `ig` of this workgroup's slab sequence (all its tiles, K-slab by K-slab) is the next one
        // to fetch, into stage ig & 1.  Consumers are released into slab g by barrier B(g); after B(g) the stage of slab g-1
        // is free, so slab g+1 may be fetched.  At a tile boundary the consumers signal "done with the tile's last slab" with
        // one extra barrier X before their epilogue, whose scratch is that slab's stage: the next tile's FIRST slab (fetched after
        // B of the last slab, into the other stage) lands while the epilogue runs; its second slab follows B of the first, when
        // the scratch is free again.
        int ig = 0, it = t_begin + (bid >> 3), ikt = 0, im0 = 0, in0 = 0, itn = 0;
        if (it < t_end) tile_origin(it, im0, in0);
        if constexpr (EP == 2) {
            if (pw == 1) {        // q / k RMSNorm weights (96 floats each), once
                const unsigned off = (unsigned)lane * 16 < 368u ? (unsigned)lane * 16 : 368u;
                glds16_s(off, (const unsigned char*)p.rms_wq, lds_off + AUX_RMS);
                glds16_s(off, (const unsigned char*)p.rms_wk, lds_off + AUX_RMS + 1024);
            }
        }
        auto issue_next = [&]() {
            if (it >= t_end) return;
            if constexpr (EP != 0) {
                if (ikt == 0) {   // the tile's per-column vectors ride with its first slab
                    issue_colp(in0, itn);
                    ++itn;
                }
            }
            issue(ig & 1, im0, in0, ikt);
            ++ig;
            if (++ikt == nk) {
                ikt = 0;
                it += stride;
                if (it < t_end) tile_origin(it, im0, in0);
            }
        };
        if constexpr (RING == 1) {
            // Two cursors over this workgroup's slab sequence (all its tiles, slab by slab): A runs two slabs ahead of the consumers, W
            // one.  Program order of this wave's DMA:  A(0) W(0) A(1) | B(0) | W(1) A(2) | B(1) | W(2) A(3) | ...  Before B(g) everything
            // but the YOUNGEST A image (A(g+1), PA / NPW pieces per plane) must have landed: a counted vmcnt, the pieces of a tile's
            // column vectors are older than the W image they precede.
            struct Cur { int t, kt, m0, n0, g; };
            Cur ca{t_begin + (bid >> 3), 0, 0, 0, 0}, cw = ca;
            if (ca.t < t_end) { tile_origin(ca.t, ca.m0, ca.n0); cw.m0 = ca.m0; cw.n0 = ca.n0; }
            auto advance = [&](Cur& c) {
                ++c.g;
                if (++c.kt == nk) { c.kt = 0; c.t += stride; if (c.t < t_end) tile_origin(c.t, c.m0, c.n0); }
            };
            int ga = 0, gw = 0, tiles_w = 0;                       // ring positions of the NEXT A / W image to fetch
            auto fetch_a = [&]() { if (ca.t < t_end) { issue_a(ga, ca.m0, ca.kt); ga = ga == 2 ? 0 : ga + 1; advance(ca); } };
            auto fetch_w = [&]() {
                if (cw.t < t_end) {
                    if (cw.kt == 0) {
                        issue_colp(cw.n0, tiles_w);
                        if constexpr (LNF) load_row_stats(cw.m0);      // one slab ahead of the consumers: the wait for these loads is off their path
                        ++tiles_w;
                    }
                    issue_w(gw, cw.n0, cw.kt); gw ^= 1; advance(cw);
                }
            };
            constexpr int YOUNG = NS * (PA / NPW);                 // pieces of one A image issued by this wave
            fetch_a(); fetch_w(); fetch_a();
            int g = 0;
            for (int t = t_begin + (bid >> 3); t < t_end; t += stride) {
                for (int kt = 0; kt < nk; ++kt, ++g) {
                    if (ca.g > g + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG) : "memory");     // A(g+1) may still fly
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pc_barrier();                      // B(g): A(g), W(g) landed; the images of slab g-1 are dead
                    fetch_w();                         // W(g+1), first: the consumers need it at B(g+1)
                    fetch_a();                         // A(g+2)
                }
                // (e4m3 class, fp32-residual epilogue: the NEXT tile's column scales are read inside this tile's epilogue -- they rode in
                // front of W(g) of that tile's first slab, fetched one slab ago; everything else waits at the next B, after the epilogue)
                if constexpr (sizeof(T) == 1 && EP == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pc_barrier();                          // X: every consumer is done with the tile's last slab
            }
            return;
        }
        issue_next();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int g = 0;
        for (int t = t_begin + (bid >> 3); t < t_end; t += stride) {
            for (int kt = 0; kt < nk; ++kt) {
                BD_PROBE_IF(g < 20, g * 3)
                pc_barrier();                          // B(g): slab g landed; slab g-1 is dead
                BD_PROBE_IF(g < 20, g * 3 + 1)
                if (ig == g + 1) issue_next();         // slab g+1 (unless it was fetched at the tile boundary already)
                BD_PROBE_IF(g < 20, g * 3 + 2)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ++g;
            }
            pc_barrier();                              // X: every consumer is done with the tile's last slab
        }
#ifdef BD_GEMM_PROBE
        if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wid / WN, wn = wid % WN;
    const int lrow = lane & 31, lhalf = lane >> 5;
#ifdef BD_GEMM_PROBE
    unsigned probe_ts = 0;
#endif
    BD_PROBE(58) BD_PROBE_RT(56)
    int g = 0;       // (the host launches this kernel only when the wide, 16-byte epilogue applies: wide_epilogue_ok)
    int ga = 0;      // RING 1: g % 3, the A image of the current slab
    int ti = 0;      // tiles done by this workgroup (parity = side-buffer slot of the tile's column vectors)
    f32x16 acc[MI][NI];
    if constexpr (EP != 0) {      // first tile; later tiles are initialised inside the previous tile's epilogue
        int m0, n0;
        tile_origin(t_begin + (bid >> 3), m0, n0);
        acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    }
    for (int t = t_begin + (bid >> 3); t < t_end; t += stride, ++ti) {
        int m0, n0;
        tile_origin(t, m0, n0);
        if constexpr (EP == 0) acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
        for (int kt = 0; kt < nk; ++kt, ++g) {
            BD_PROBE_IF(g < 20, g * 3)
            pc_barrier();                              // B(kt)
            BD_PROBE_IF(g < 20, g * 3 + 1)
            const unsigned char* const baseA = lds + off_a(RING ? ga : (g & 1));
            const unsigned char* const baseW = lds + off_w(RING ? (g & 1) : (g & 1));
            if constexpr (RING == 1) ga = ga == 2 ? 0 : ga + 1;
            constexpr int FB = NS == 1 ? 2 : 1;
            frag_t a[FB][NS][MI], b[FB][NS][NI];
#define LOAD_ONE(dst, ptr, row, ks)                                                                           \
            {                                                                                                 \
                const int c0_ = ((ks) * 2 + lhalf) * CPF;                                                     \
                if constexpr (CPF == 1) {                                                                     \
                    dst = __builtin_bit_cast(frag_t, *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4))); \
                } else {                                                                                      \
                    const u128 lo_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4)); \
                    const u128 hi_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_ + 1) << 4)); \
                    dst = (frag_t){(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
                }                                                                                             \
            }
#define LOAD_FRAGS(ks, slot)                                                                                  \
            _Pragma("unroll") for (int s2 = 0; s2 < NS; ++s2) {                                               \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                \
                    LOAD_ONE(a[slot][s2][i], baseA + s2 * A_BYTES, wm * (MI * 32) + i * 32 + lrow, ks)         \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                \
                    LOAD_ONE(b[slot][s2][j], baseW + s2 * W_BYTES, wn * (NI * 32) + j * 32 + lrow, ks) \
            }
            if constexpr (NS == 1) {
                // Software pipeline inside the slab, order PINNED (sched_barrier(0) after every MFMA / ds_read pair): the
                // fragment reads of k-step ks+1 are interleaved one-for-one with the MFMAs of k-step ks, so a consumer wave
                // keeps its matrix pipe fed on its own.  (With producers doing the DMA, both consumer waves of a SIMD leave the
                // slab barrier at the same instant; left to itself hipcc re-associates the unrolled k-steps into chains of
                // dependent MFMAs on one accumulator, each behind an lgkmcnt(0) -- 3100-3600 cycles per slab for 2048 cycles of
                // matrix work per SIMD, profiles/r2_gemm_phase_probe.md.)  MFMA q of a k-step works on tile (q % MI, q / MI);
                // reads go a0, b0, a1, b1, ... so that the operands the next k-step needs first arrive first; hipcc keeps the
                // lgkmcnt waits counted (its own ds_reads, in order).
                constexpr int NRD = MI + NI, NMM = MI * NI;
                auto load_q = [&](int slot, int ks, int q) {
                    const int which = (q < 2 * (MI < NI ? MI : NI)) ? (q & 1) : (MI > NI ? 0 : 1);   // 0: an A fragment, 1: a W fragment
                    const int idx = (q < 2 * (MI < NI ? MI : NI)) ? (q >> 1) : (q - (MI < NI ? MI : NI));
                    if (which == 0) LOAD_ONE(a[slot][0][idx], baseA, wm * (MI * 32) + idx * 32 + lrow, ks)
                    else LOAD_ONE(b[slot][0][idx], baseW, wn * (NI * 32) + idx * 32 + lrow, ks)
                };
#pragma unroll
                for (int q = 0; q < NRD; ++q) load_q(0, 0, q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int cur = ks & 1;
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(a[cur][0][i], b[cur][0][j], acc[i][j]);
                        if (ks + 1 < KS && q < NRD) load_q(cur ^ 1, ks + 1, q);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // Split-bf16 (NS == 2): per k-step and output tile three MFMAs -- lo(A) hi(W), hi(A) lo(W), hi(A) hi(W) -- on ten
                // fragments (A hi/lo x MI, W hi/lo x NI = 40 VGPRs).  Order pinned so that registers rotate without a second
                // full fragment set (96 accumulator + 60 fragment registers fit the 168-VGPR budget):
                //   phase 1  lo(A) * hi(W)   -- afterwards the lo(A) registers are dead: the NEXT k-step's lo(A) loads go there
                //   phase 2  hi(A) * lo(W)   -- lo(A) loads in flight; afterwards lo(W) is dead: next lo(W) loads go there
                //   phase 3  hi(A) * hi(W)   -- next lo(W), then next hi(W), hi(A) loads (into the spare hi set) in flight
                // so one ds_read rides under almost every MFMA and no wave depends on its SIMD partner to cover its reads.
                constexpr int NMM = MI * NI;
                frag_t ah[2][MI], wh[2][NI], al[MI], wl[NI];
                const unsigned char* wbase = baseW;
#define LD_A(dst, plane, idx, ks) LOAD_ONE(dst, baseA + (plane) * A_BYTES, wm * (MI * 32) + (idx) * 32 + lrow, ks)
#define LD_W(dst, plane, idx, ks) LOAD_ONE(dst, wbase + (plane) * W_BYTES, wn * (NI * 32) + (idx) * 32 + lrow, ks)
#pragma unroll
                for (int j = 0; j < NI; ++j) LD_W(wh[0][j], 0, j, 0)
#pragma unroll
                for (int i = 0; i < MI; ++i) LD_A(al[i], 1, i, 0)
#pragma unroll
                for (int i = 0; i < MI; ++i) LD_A(ah[0][i], 0, i, 0)
#pragma unroll
                for (int j = 0; j < NI; ++j) LD_W(wl[j], 1, j, 0)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int cur = ks & 1, nxt = cur ^ 1;
                    const bool more = ks + 1 < KS;
                    // phase 1: lo(A) * hi(W)
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(al[i], wh[cur][j], acc[i][j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // phase 2: hi(A) * lo(W), next lo(A) loads riding under the first MFMAs
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(ah[cur][i], wl[j], acc[i][j]);
                        if (more && q < MI) LD_A(al[q], 1, q, ks + 1)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // phase 3: hi(A) * hi(W), next lo(W), hi(W), hi(A) loads under it
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(ah[cur][i], wh[cur][j], acc[i][j]);
                        if (more) {
                            if (q < NI) LD_W(wl[q], 1, q, ks + 1)
                            else if (q < 2 * NI) LD_W(wh[nxt][q - NI], 0, q - NI, ks + 1)
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (more) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) LD_A(ah[nxt][i], 0, i, ks + 1)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#undef LD_A
#undef LD_W
            }
#undef LOAD_FRAGS
#undef LOAD_ONE
        }
        BD_PROBE_IF(g == nk, 60)
        pc_barrier();                                  // X: this wave is done reading the tile's last slab
        BD_PROBE_IF(g == nk, 61)
        {   // LDS-staged epilogue; its scratch is the stage the tile's last slab lived in (free after X)
            unsigned char* scratch = lds + ((g - 1) & 1) * STAGE_BYTES + wid * (SR * NI * 32 * 4);
            unsigned char* scratch_hi = scratch + 8 * NI * 32 * 4;
            if constexpr (RING == 1) {         // the W and the A image of the tile's last slab (ga already points one past it)
                scratch = lds + off_w((g - 1) & 1) + wid * (8 * NI * 32 * 4);
                scratch_hi = lds + off_a(ga == 0 ? 2 : ga - 1) + wid * (8 * NI * 32 * 4);
            }
            if constexpr (EP == 0) {
                gemm_epilogue_lds<T, NS, MI, NI, SR, true>(p, acc, scratch, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
            } else {
                const bool has_next = t + stride < t_end;
                int nm0 = 0, nn0 = 0;
                if (has_next) tile_origin(t + stride, nm0, nn0);
                pc_epilogue<T, NS, EP, OUTK, GELU, MI, LNF>(p, acc, (float*)scratch, (float*)scratch_hi, (const float*)(lds + AUX_COLP + (ti & 1) * 2048),
                                                  (const float*)(lds + AUX_COLP + ((ti + 1) & 1) * 2048), (const float*)(lds + AUX_RMS), wn * (NI * 32), m0 + wm * (MI * 32), n0 + wn * (NI * 32),
                                                  lane, has_next, nm0 + wm * (MI * 32), nn0 + wn * (NI * 32),
                                                  (const float*)(lds + (LNF ? AUX_ROWS : 0) + (ti & 1) * 2048), wm * (MI * 32));
            }
        }
        BD_PROBE_IF(g == nk, 62)
    }
    BD_PROBE(59) BD_PROBE_RT(57)
#ifdef BD_GEMM_PROBE
    if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
}

// Which epilogue specialisation of gemm_kernel_pc serves these arguments (0 = the generic one): see the kernel's header
template <class T, int NS> int pc_epilogue_kind(const bd_gemm_args& a, int& outk, bool& gelu) {
    outk = a.out_f32;
    gelu = a.act == BD_ACT_GELU;
    if (a.addtab || a.rpg_in > 0) return 0;
    if (a.bias && ((uintptr_t)a.bias & 15)) return 0;
    if (a.out_f32 == OUT_F32) {
        if (gelu || a.rms_wq) return 0;
        if (a.wscale && (sizeof(T) != 1 || ((uintptr_t)a.wscale & 15))) return 0;
        return 3;
    }
    if (a.resid) return 0;
    if (a.wscale && (sizeof(T) != 1 || ((uintptr_t)a.wscale & 15))) return 0;
    // output kinds instantiated per operand class: native everywhere; f16 from split-bf16 (strict f16 attention), bf16 from e4m3
    const bool outk_ok = outk == OUT_OPERAND || (outk == OUT_F16 && NS == 2 && sizeof(T) == 2) || (outk == OUT_BF16 && sizeof(T) == 1) ||
                         (outk == OUT_BF16X2 && NS == 2 && std::is_same<T, _Float16>::value);   // split-f16 QKV feeding split-bf16 attention
    if (!outk_ok) return 0;
    if (a.rms_wq) return gelu ? 0 : 2;
    if (gelu && outk != OUT_OPERAND) return 0;
    return 1;
}

template <class T, int NS, int BK, int WM, int WN, int MI, int NI> void launch_pc(const bd_gemm_args& a, hipStream_t s, int cus) {
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    constexpr int NPW = 4;        // one producer wave per SIMD: a single wave issues one LDS-DMA piece per ~70 cycles, the CU ~23
    const int tiles = ((a.M + TBM - 1) / TBM) * ((a.N + TBN - 1) / TBN);
    const int grid = tiles < cus ? tiles : cus;
    const dim3 g(grid), b((WM * WN + NPW) * 64);
    int outk = 0;
    bool gelu = false;
    const int ep = (a.N % TBN == 0) ? pc_epilogue_kind<T, NS>(a, outk, gelu) : 0;
    // Operand ring: the asymmetric A3 / W2 form for every specialised epilogue (same-box A/B, profiles/r4_gemm_ring.md: K = 3072 +8-16 %,
    // split classes +3-9 %, e4m3 +4 %, K = 768 within 1 %) except the shallow fp32-residual Linears of the plain 16-bit classes (proj,
    // K = 768: HBM-bound, 2.5-5 % slower with it).  A row's result does not depend on the choice (same K order): it follows the
    // launch's class, K and epilogue only, never M.
    const bool ring0_ep3 = NS == 1 && sizeof(T) == 2 && a.K < 1536;
#define BD_PC_LAUNCH_R(EP_, OUTK_, GELU_, RING_) hipLaunchKernelGGL((gemm_kernel_pc<T, NS, BK, WM, WN, MI, NI, NPW, EP_, OUTK_, GELU_, RING_>), g, b, 0, s, a)
#define BD_PC_LAUNCH(EP_, OUTK_, GELU_) BD_PC_LAUNCH_R(EP_, OUTK_, GELU_, ((EP_) != 0 ? 1 : 0))
    constexpr int ALT = (NS == 2 && sizeof(T) == 2) ? OUT_F16 : (sizeof(T) == 1 ? OUT_BF16 : OUT_OPERAND);   // the one non-native 16-bit kind
    constexpr bool X2 = NS == 2 && std::is_same<T, _Float16>::value;     // split-f16 also emits split-bf16 planes (attention input)
    if (ln_fold_consumer(a)) {          // LayerNorm fold, consumer side: the f16 q, k launch of the default mode (bd_gemm_takes_ln_fold)
        if constexpr (NS == 1 && std::is_same<T, _Float16>::value)
            hipLaunchKernelGGL((gemm_kernel_pc<T, NS, BK, WM, WN, MI, NI, NPW, 2, OUT_OPERAND, false, 1, true>), g, b, 0, s, a);
    }
    else if (ep == 3 && ring0_ep3) BD_PC_LAUNCH_R(3, OUT_F32, false, 0);
    else if (ep == 3) BD_PC_LAUNCH(3, OUT_F32, false);
    else if (ep == 2 && outk == OUT_BF16X2) { if constexpr (X2) BD_PC_LAUNCH(2, OUT_BF16X2, false); }
    else if (ep == 2) { if (outk == OUT_OPERAND) BD_PC_LAUNCH(2, OUT_OPERAND, false); else BD_PC_LAUNCH(2, ALT, false); }
    else if (ep == 1 && gelu && outk == OUT_OPERAND) BD_PC_LAUNCH(1, OUT_OPERAND, true);
    else if (ep == 1 && !gelu && outk == OUT_BF16X2) { if constexpr (X2) BD_PC_LAUNCH(1, OUT_BF16X2, false); }
    else if (ep == 1 && !gelu) { if (outk == OUT_OPERAND) BD_PC_LAUNCH(1, OUT_OPERAND, false); else BD_PC_LAUNCH(1, ALT, false); }
    else BD_PC_LAUNCH(0, OUT_OPERAND, false);
#undef BD_PC_LAUNCH
#undef BD_PC_LAUNCH_R
}

template <class T, int NS, int BK, int WM, int WN, int MI, int NI, int NSTG = 2> void launch_glds(const bd_gemm_args& a, hipStream_t s) {
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    const int tiles = ((a.M + TBM - 1) / TBM) * ((a.N + TBN - 1) / TBN);
    hipLaunchKernelGGL((gemm_kernel_glds<T, NS, BK, WM, WN, MI, NI, NSTG>), dim3(tiles), dim3(WM * WN * 64), 0, s, a);
}

template <class T> bd_gemm_args row_slice(const bd_gemm_args& a, int64_t row0, int rows) {
    bd_gemm_args sub = a;
    constexpr int ESZ = OpGeom<T>::ESZ;
    const int osz = a.out_f32 == OUT_F32 ? 4 : (a.out_f32 == OUT_OPERAND ? ESZ : 2);
    sub.A = (const unsigned char*)a.A + row0 * a.lda * ESZ;
    sub.out = (unsigned char*)a.out + row0 * a.ldo * osz;
    if (a.resid) sub.resid = a.resid + row0 * a.ldr;
    sub.M = rows;
    return sub;
}

// does this problem go to the persistent 256 x 192 kernel (every wave tile = 96 consecutive output columns)?
inline bool pc192_possible(const bd_gemm_args& a, int ns, int esz) {
    return wide_epilogue_ok(a, ns) && 256 * a.lda * esz < ((int64_t)1 << 31) && 256 * a.ldw * esz < ((int64_t)1 << 31);
}
inline bool uses_pc192(const bd_gemm_args& a, int ns, int esz, int cus) {
    if (!pc192_possible(a, ns, esz)) return false;
    if (a.rms_wq) return true;       // a fused q/k RMSNorm pins the tile shape for EVERY batch size: a sample's q, k must not
                                     // depend on whether its batch filled the CUs (bit-exact batch independence is tested)
    if (a.N % 192 || a.M < 1024) return false;
    // (sub-batch lanes enqueue this launch once per lane, side by side: the round their tiles fill is the round of all of them)
    const int64_t t192 = (int64_t)((a.M + 255) / 256) * (a.N / 192) * bd_concurrent_launches();
    // (0.75: DINOv2's N = 768 GEMMs at M = 50112 are 784 tiles = 3.06 rounds, fill 0.77.  One batch at a time the persistent kernel
    // is then +1.8 % on the step against the one-tile kernels' hybrid split; with two batches in flight -- the other batch's kernels
    // run on the CUs this kernel's tail leaves idle -- +5 %: 1221 -> 1282 poses/s same box.)
    // Round 5 re-measured the threshold over batch sizes 2 ... 32 with the lanes' tiles counted together (profiles/r5_small_experiments.md):
    // 0.45 is never slower than 0.75 and 5-11 % faster at batch 5, 6, 12, 14 (0.35 loses 2 % at batch 3).
    return (double)t192 / (double)(((t192 + cus - 1) / cus) * cus) >= 0.45;       // occupancy of the CUs over the launch's rounds
}
// a fused q/k RMSNorm needs: 16-bit output of a plain Linear, N = 3 x heads x 96, row-identity output map, and N a multiple of
// the 192-column workgroup tile: with an odd head count (N = 288 h, h odd) the last column tile's second wave tile would lie
// past N, and the fused branch stores whole 96-column heads without a column guard

template <class T, int NS, int BK> void launch_one_tile(const bd_gemm_args& a, hipStream_t s, int kCUs);

// LayerNorm fold in the classes of this file: the consumer side only, and only the f16 launch with the fused q/k RMSNorm (which pins the
// persistent 256 x 192 kernel for every row count)
template <class T, int NS> bool takes_ln_fold(const bd_gemm_args& a) {
    if (ln_fold_producer(a)) return false;
    if (!ln_fold_consumer(a)) return true;
    if (!(NS == 1 && std::is_same<T, _Float16>::value)) return false;
    int outk = 0;
    bool gelu = false;
    return ln_fold_consumer_ok(a) && a.rms_wq && rms_geometry_ok(a) && pc192_possible(a, NS, OpGeom<T>::ESZ) && a.out_f32 == OUT_OPERAND &&
           pc_epilogue_kind<T, NS>(a, outk, gelu) == 2;
}

template <class T, int NS, int BK> int launch(const bd_gemm_args& a, hipStream_t s) {
    const int kCUs = cu_count();
    if (a.rms_wq && !(rms_geometry_ok(a) && pc192_possible(a, NS, OpGeom<T>::ESZ))) return BD_ERR_SHAPE;
    if ((ln_fold_producer(a) || ln_fold_consumer(a)) && !takes_ln_fold<T, NS>(a)) return BD_ERR_SHAPE;
    const int slot = bd_trace_open(s, 0, a.M, a.N, a.K);
    constexpr int ESZ_ = OpGeom<T>::ESZ;
    // 256 x 192 tiles (8 consumer waves of 64 x 96: 96 accumulator + 2 x 20 fragment registers fit the 168-VGPR budget
    // of three waves per SIMD with the fragment double-buffering intact) whenever they tile N exactly -- every Linear of
    // both stacks except the head (N = 2304, 3072, 768 are multiples of 192).
    if (uses_pc192(a, NS, ESZ_, kCUs)) {
        const int64_t rows_main = pc192_main_rows(a, kCUs);
        if (rows_main < a.M) {
            launch_pc<T, NS, BK, 4, 2, 2, 3>(row_slice<T>(a, 0, (int)rows_main), s, kCUs);
            launch_one_tile<T, NS, BK>(row_slice<T>(a, rows_main, (int)(a.M - rows_main)), s, kCUs);
        } else
            launch_pc<T, NS, BK, 4, 2, 2, 3>(a, s, kCUs);
    } else
        launch_one_tile<T, NS, BK>(a, s, kCUs);
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

// the one-tile-per-workgroup kernels: everything the persistent kernel does not take
template <class T, int NS, int BK> void launch_one_tile(const bd_gemm_args& a, hipStream_t s, int kCUs) {
    {
        // Tile choice = best estimated efficiency: wave quantisation over the resident slots (256x256: one workgroup
        // per CU; 128x128: two; 64x64: four) times the measured relative mainloop efficiency of the tile
        // (profiles/r1_gemm_experiments.md: 128x128 ~0.87 of 256x256 on the wide shapes; 64x64 ~0.55).
        auto eff = [&](int tm, int tn, int slots, double base) {
            const double tiles = (double)((a.M + tm - 1) / tm) * ((a.N + tn - 1) / tn);
            const double rounds = (double)(((int64_t)tiles + slots - 1) / slots);
            return base * tiles / (rounds * slots);
        };
        const double e256 = a.N >= 1536 ? eff(256, 256, kCUs, 1.0) : 0.0;
        const double e128 = eff(128, 128, 2 * kCUs, 0.87);
        const double e64 = eff(64, 64, 4 * kCUs, 0.55);
        // Narrow outputs (N = 768: proj / fc2) have too few 256x256 tiles for whole rounds (576 tiles = 2.25 rounds) and
        // pay 4.5 -> 5 rounds plus the weaker mainloop with 128x128 tiles.  Hybrid: k FULL rounds of 256x256 tiles over
        // the first rows, the remaining rows as 128x128 tiles -- two launches over disjoint row ranges, no split-K, no
        // cross-workgroup fix-up, bit-identical per-row arithmetic order (the K order of a row does not depend on the
        // tile).  Cost model in units of one 256x256 round; a 128x128 round is 2 tiles per CU at 0.87 efficiency.
        // Only for deep K (fc2: 324 -> 297 us): at K = 768 (proj) the GEMM is bound by its fp32 residual epilogue, which
        // two co-resident 128x128 workgroups overlap better than one 256x256 workgroup (117 vs 125 us).
        int k256 = 0;
        int64_t rows256 = 0;
        double hybrid_cost = 1e30;
        if (a.N % 256 == 0 && a.N < 1536 && a.K >= 2048 && a.rpg_in <= 0 && !a.addtab && a.M >= 1024) {
            const int tn = a.N / 256;
            const int64_t mtiles = (a.M + 255) / 256;
            auto cost128 = [&](int64_t rows) {
                if (rows <= 0) return 0.0;
                const int64_t t = ((rows + 127) / 128) * (a.N / 128);
                const int64_t full = t / (2 * kCUs), rem = t % (2 * kCUs);
                // a partial last round runs with one workgroup on most CUs: faster than a full round
                return full * 0.575 + (rem == 0 ? 0.0 : (rem <= kCUs ? 0.36 : 0.575));
            };
            double best = cost128(a.M);
            for (int k = 1; k <= 64; ++k) {
                const int64_t mt = (int64_t)k * kCUs / tn;
                if (mt > mtiles) break;
                const int64_t r256 = mt * 256 < a.M ? mt * 256 : a.M;
                const double c = k + cost128(a.M - r256);
                if (c < best - 1e-9) { best = c; k256 = k; rows256 = r256; }
            }
            hybrid_cost = best;
        }
        // 256 x 192 tiles (8 waves of 64 x 96): N = 768 in 4 columns of tiles, 192 row tiles x 4 = exactly 3 rounds at
        // M = 49152 (BETR fc2) -- whole rounds of a big tile without any row split.
        bool use192 = false;
        if (a.N % 192 == 0 && a.N < 1536 && a.K >= 2048 && a.M >= 1024) {
            const double e192 = eff(256, 192, kCUs, 0.93);
            const double ehyb = k256 > 0 ? (double)((a.M + 255) / 256) * (a.N / 256) / kCUs / hybrid_cost : e128;
            use192 = e192 > ehyb && e192 > e128;
        }
        if constexpr (NS == 1) {
            if (use192) {
                launch_glds<T, NS, BK, 4, 2, 2, 3>(a, s);
                return;
            }
        }
        if (k256 > 0) {
            const bd_gemm_args big = row_slice<T>(a, 0, (int)rows256);
            launch_glds<T, NS, BK, 2, 4, 4, 2>(big, s);
            if (rows256 < a.M) {
                const bd_gemm_args rest = row_slice<T>(a, rows256, (int)(a.M - rows256));
                launch_glds<T, NS, BK, 2, 2, 2, 2>(rest, s);
            }
        } else if (e256 >= e128 && e256 >= e64)
            launch_glds<T, NS, BK, 2, 4, 4, 2>(a, s);             // 256 x 256, one tile per workgroup
        else {
            // Sparse launches (small batch: every tile gets its own workgroup slot at once, so the launch lasts one tile's chain of
            // slab latencies): the 4-stage ring form of the same tile (three slabs in flight; 128 / 64 KiB of LDS: one / two per CU).
            const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128), t64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
            if (e128 >= e64) {
                if (t128 <= kCUs) launch_glds<T, NS, BK, 2, 2, 2, 2, 4>(a, s);
                else launch_glds<T, NS, BK, 2, 2, 2, 2>(a, s);    // 128 x 128, 2 workgroups / CU
            } else {
                if (t64 <= 2 * kCUs) launch_glds<T, NS, BK, 2, 2, 1, 1, 4>(a, s);
                else launch_glds<T, NS, BK, 2, 2, 1, 1>(a, s);    // 64 x 64 (latency mode: batch 1, M = 1536)
            }
        }
    }
}

}  // namespace

extern "C" int bd_gemm_fuses_qk_rmsnorm(const bd_gemm_args* args, int prec) {
    if (!args || !rms_geometry_ok(*args)) return 0;
    switch (prec) {
        case BD_PREC_BF16: case BD_PREC_F16: return pc192_possible(*args, 1, 2) ? 1 : 0;
        case BD_PREC_BF16X3: case BD_PREC_F16X3: return pc192_possible(*args, 2, 2) ? 1 : 0;
        case BD_PREC_FP8: return pc192_possible(*args, 1, 1) ? 1 : 0;
        case BD_PREC_F16C8: return wide_epilogue_ok(*args, 2) ? 1 : 0;
        default: return 0;
    }
}

extern "C" int bd_gemm_takes_ln_fold(const bd_gemm_args* args, int prec) {
    if (!args) return 0;
    const bd_gemm_args& a = *args;
    if (!ln_fold_producer(a) && !ln_fold_consumer(a)) return 1;
    switch (prec) {
        case BD_PREC_F16: return takes_ln_fold<_Float16, 1>(a) ? 1 : 0;
        case BD_PREC_F16C8: return bd_f16c8_takes_ln_fold(a) ? 1 : 0;
        default: return 0;
    }
}

extern "C" int bd_gemm(const bd_gemm_args* args, int prec, void* stream) {
    if (!args || !args->A || !args->W || !args->out) return BD_ERR_NULL;
    const bd_gemm_args& a = *args;
    const int kmult = prec == BD_PREC_FP8 ? 128 : 64;
    const int esz = prec == BD_PREC_FP8 ? 1 : 2;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % kmult) != 0) return BD_ERR_SHAPE;
    if (((a.lda * esz) % 16) || ((a.ldw * esz) % 16) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15)) return BD_ERR_ALIGN;
    if ((prec == BD_PREC_BF16X3 || prec == BD_PREC_F16X3 || prec == BD_PREC_F16C8) && ((a.a_plane % 8) || (a.w_plane % 8))) return BD_ERR_ALIGN;
    if (a.addtab && a.tab_rows <= 0) return BD_ERR_SHAPE;
    if (a.out_f32 < 0 || a.out_f32 > 5) return BD_ERR_DTYPE;
    if (a.out_f32 >= 4 && (a.out_plane % 8)) return BD_ERR_ALIGN;
    // the single-plane 16-bit outputs and every fp8-mode output exist only in the wide (16-byte) epilogue
    const bool needs_wide = a.out_f32 >= 2 || (prec == BD_PREC_FP8 && a.out_f32 == 0);
    if (needs_wide && ((a.N % 8) || (a.ldo % 8) || ((uintptr_t)a.out & 15))) return BD_ERR_ALIGN;
    if (a.wscale && ((uintptr_t)a.wscale & 15)) return BD_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return launch<__bf16, 1, 64>(a, s);
        case BD_PREC_F16: return launch<_Float16, 1, 64>(a, s);
        case BD_PREC_BF16X3: return launch<__bf16, 2, 32>(a, s);
        case BD_PREC_F16X3: return launch<_Float16, 2, 32>(a, s);
        case BD_PREC_FP8: return launch<fp8e4, 1, 128>(a, s);
        case BD_PREC_F16C8: return bd_launch_gemm_f16c8(a, s);
        default: return BD_ERR_DTYPE;
    }
}

