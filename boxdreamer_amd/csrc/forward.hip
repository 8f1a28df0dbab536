// Whole-path entry points: the DINOv2 encoder (DinoV2Wrapper.predict) and the BETR decoder
// (BETR.forward) as straight-line sequences of kernel launches on the caller's stream.
// No allocation, no synchronisation, no state: the caller provides one workspace blob that is
// carved here (256-byte aligned slices).
#include "bd_common.h"

namespace {

struct Carver {
    unsigned char* base;
    size_t off;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

inline int planes_of(int prec) {      // 2-byte units per element of a 16-bit operand buffer (F16C8: f16 plane + e4m3 plane)
    return (prec == BD_PREC_BF16X3 || prec == BD_PREC_BF16X3_ATTN_X3 || prec == BD_PREC_F16C8 || prec == BD_PREC_F16C8_QK16 ||
            prec == BD_PREC_F16X3 || prec == BD_PREC_F16X3_ATTN_X3) ? 2 : 1;
}

struct BlockBufs {
    float* x;        // fp32 residual stream [M, D]
    void* xn;        // 16-bit LN output     [M, D]
    void* qkv;       // 16-bit               [M, 3D]
    void* ao;        // 16-bit attention out [M, D]
    void* h;         // 16-bit MLP hidden    [M, 4D]
    float* st;       // LayerNorm fold: (mean, M2) per row and 96-column group [M, D / 96, 2] fp32 (F16C8 family)
    void* sk;        // split-K scratch of the residual Linears (bd_gemm_args.sk_ws; small M only, else NULL); flags zeroed per forward
    size_t sk_flag_bytes;
};

inline bd_gemm_args gemm_args(const void* A, int64_t lda, int64_t a_plane, const bd_linear& lin, int64_t ldw, int N,
                              void* out, int64_t ldo, int64_t out_plane, int out_f32, int M, int K, int act) {
    bd_gemm_args g{};
    g.A = A; g.lda = lda; g.a_plane = a_plane;
    g.W = lin.w; g.ldw = ldw; g.w_plane = (int64_t)N * ldw;
    g.bias = lin.b;
    g.wscale = lin.wscale;
    g.out = out; g.ldo = ldo; g.out_plane = out_plane; g.out_f32 = out_f32;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.w_qexp = lin.w_qexp;
    return g;
}

#define BD_TRY(expr) do { int rc__ = (expr); if (rc__ != BD_OK) return rc__; } while (0)

// Attention policy of the strict (split-bf16 x3) family.  The variants are separate `prec` values of the whole-path
// entry points (no environment switches, no library state):
//   BD_PREC_BF16X3           GEMMs split-bf16; attention as ONE f16 pass where q and k are RMS-normalised (BETR: the x3 QKV
//                            GEMM stores q, k, v as a single f16 plane, q/k RMSNorm and attention run in f16, attention writes
//                            (hi, lo) bf16 planes for the x3 proj GEMM); DINOv2's attention (un-normalised q.k) stays
//                            split-bf16.  Measured 1.3e-4 on the logits at full depth, T = 6.
//   BD_PREC_BF16X3_ATTN_X3   split-bf16 attention everywhere (8.2e-5, ~8 % slower)
// (f16 attention everywhere -- ABI <= 6's BD_PREC_BF16X3_ATTN_F16 -- measured 1.18e-3: the f16 Q.K^T on DINOv2's un-normalised
// q / k eats the whole budget; removed in ABI 7.)
inline int gemm_prec(int prec) {
    if (prec == BD_PREC_F16C8_QK16) return BD_PREC_F16C8;
    if (prec == BD_PREC_F16X3_ATTN_X3) return BD_PREC_F16X3;
    return prec == BD_PREC_BF16X3_ATTN_X3 ? BD_PREC_BF16X3 : prec;
}
inline bool qk_single_f16(int prec) { return prec == BD_PREC_F16C8_QK16; }
inline bool x3_f16_attention(int prec, bool qk_normed) { return (prec == BD_PREC_BF16X3 || prec == BD_PREC_F16X3) && qk_normed; }
inline bool split16(int cls) { return cls == BD_PREC_BF16X3 || cls == BD_PREC_F16X3; }
// operand class of a Linear whose BD_PROMOTE_* bit is set (include/boxdreamer_hip.h): F16C8 family -> split-f16, e4m3 -> bf16
inline int promoted_class(int base) { return base == BD_PREC_F16C8 ? BD_PREC_F16X3 : (base == BD_PREC_FP8 ? BD_PREC_BF16 : base); }
inline int lin_class(int base, int promote, int bit) { return (promote & bit) ? promoted_class(base) : base; }
// fc1 -> fc2 and adapter fc1 -> fc2 hand-offs: a promoted GEMM cannot emit the base operand class, so promoting the first promotes the second
inline int norm_promote(int pm) { return (pm & BD_PROMOTE_FC1) ? (pm | BD_PROMOTE_FC2) : pm; }
// output kind (bd_gemm_args.out_f32) with which a GEMM of class `from` writes the A operand of a Linear of class `to`
inline int handoff_kind(int from, int to) {
    return from == to ? 0 : (to == BD_PREC_BF16 ? 3 /* e4m3 GEMM -> bf16 plane */ : 5 /* F16C8 GEMM -> split-f16 planes */);
}

// How one transformer block runs: the operand class of each Linear (F16C8 family: per-Linear promotion to split-f16; e4m3: to bf16), the attention
// form, and the 16-bit kinds in which producers hand their results on.
struct BlockPlan {
    int base;                        // operand class of the un-promoted Linears
    int c_qkv, c_proj, c_fc1, c_fc2; // operand class per Linear
    bool hyb;                        // attention as ONE f16 pass on a single f16 q, k, v plane (q, k RMS-normalised)
    bool qk16;                       // BETR's QKV Linear split by column (q, k one f16 pass on the f16 plane, v the full F16C8 product)
    int qkv_out;                     // output kind of the QKV GEMM
    int aprec_in;                    // class of the attention input (for the stand-alone q/k RMSNorm)
    int aprec;                       // bd_attention precision code (input form x class of proj's A operand)
};
inline BlockPlan plan_block(const bd_block_weights& w, int wprec) {
    BlockPlan p{};
    p.base = gemm_prec(wprec);
    const bool c8 = p.base == BD_PREC_F16C8, f8 = p.base == BD_PREC_FP8, normed = w.q_norm_w != nullptr;
    const int pm = (c8 || f8) ? norm_promote(w.promote) : 0;
    p.c_qkv = lin_class(p.base, pm, BD_PROMOTE_QKV);
    p.c_proj = lin_class(p.base, pm, BD_PROMOTE_PROJ);
    p.c_fc1 = lin_class(p.base, pm, BD_PROMOTE_FC1);
    p.c_fc2 = lin_class(p.base, pm, BD_PROMOTE_FC2);
    p.hyb = (split16(p.base) && x3_f16_attention(wprec, normed)) || (c8 && normed && !(pm & BD_PROMOTE_ATTN));
    const bool plain_qkv = p.c_qkv == p.base;          // the special QKV forms exist for the un-promoted Linear only
    p.qk16 = qk_single_f16(wprec) && p.hyb && w.qkv16.w && plain_qkv;
    if (p.hyb) { p.qkv_out = 2; p.aprec_in = BD_PREC_F16; }                                  // one f16 plane
    else if (f8) { p.qkv_out = p.c_qkv == BD_PREC_FP8 ? 3 : 0; p.aprec_in = BD_PREC_BF16; }   // one bf16 plane (an e4m3 or a bf16 GEMM's)
    else if (p.c_qkv == BD_PREC_F16C8 || p.c_qkv == BD_PREC_F16X3) { p.qkv_out = 4; p.aprec_in = BD_PREC_BF16X3; }   // split-bf16 planes for
                                                                                              // the split-bf16 attention (range: probabilities)
    else { p.qkv_out = 0; p.aprec_in = p.c_qkv; }                                             // the GEMM's own operand class
    if (p.hyb) p.aprec = p.c_proj == BD_PREC_F16C8 ? BD_PREC_F16_OUT_F16C8 : (p.c_proj == BD_PREC_F16X3 ? BD_PREC_F16_OUT_F16X3 : BD_PREC_F16_OUT_BF16X3);
    else if (f8) p.aprec = p.c_proj == BD_PREC_FP8 ? BD_PREC_BF16_OUT_FP8 : BD_PREC_BF16;
    else if (p.aprec_in == BD_PREC_BF16X3)
        p.aprec = p.c_proj == BD_PREC_F16C8 ? BD_PREC_BF16X3_OUT_F16C8 : (p.c_proj == BD_PREC_F16X3 ? BD_PREC_BF16X3_OUT_F16X3 : BD_PREC_BF16X3);
    else p.aprec = p.base;
    return p;
}

// ---- LayerNorm fold (ABI 8, include/boxdreamer_hip.h bd_gemm_args.ln_*): between a residual Linear (proj, fc2) and the Linear(s) behind the
// next LayerNorm (fc1; the next block's QKV) of the F16C8 family the LayerNorm launch is replaced by (a) the residual Linear's epilogue
// emitting the raw row as the F16C8 operand + per-wave-tile (mean, M2) pairs and (b) the consumer's epilogue applying the row statistics to a
// product with the gain-folded weight (bd_block_weights.qkv_f / fc1_f / qkv16_f).  A hand-off folds only when EVERY launch on both sides has a
// kernel form for it (bd_gemm_takes_ln_fold) -- un-promoted F16C8 Linears, D = 768; everything else keeps the bd_layernorm launch.  The
// first LayerNorm of a stack (no residual Linear in front of it) and the stacks' final norms stay kernels.
inline void ln_consumer(bd_gemm_args& g, const BlockBufs& b, const float* colsum, float eps) { g.ln_stats_in = b.st; g.ln_colsum = colsum; g.ln_eps = eps; }
inline void ln_producer(bd_gemm_args& g, const BlockBufs& b, int64_t plane, int D) { g.ln_stats_out = b.st; g.ln_op_out = b.xn; g.ln_op_plane = plane; g.ln_op_ld = D; }
// launch, or (check) only ask whether the launch's kernel form takes the fold fields that are set
inline int gemm_or_check(bd_gemm_args& g, int cls, void* stream, bool check) {
    if (check) return bd_gemm_takes_ln_fold(&g, cls) ? BD_OK : BD_ERR_SHAPE;
    return bd_gemm(&g, cls, stream);
}

// LayerNorm 1 + QKV Linear (+ q/k RMSNorm) of a block on M rows: leaves q, k, v in b.qkv in the form the plan's attention reads.
// BD_PREC_F16C8_QK16: the QKV Linear of a block whose q, k are RMS-normalised, split by output column (include/boxdreamer_hip.h):
// LayerNorm 1 emits the F16C8 operand; launch 1 multiplies its f16 plane with the f16 copy of the q, k weight rows (one MFMA pass,
// q/k RMSNorm fused where the launch allows it), launch 2 is the full F16C8 product for the v rows.  q, k, v land in one f16
// [M, 3D] buffer exactly as the single-launch forms lay them out.
// folded: LayerNorm 1 is folded -- b.xn / b.st already hold the raw operand copy and the row statistics of b.x (the previous block's fc2
// wrote them); check: launch nothing, return BD_OK iff every launch of the folded form has a kernel form.
int qkv_stage(const bd_block_weights& w, const BlockPlan& p, const BlockBufs& b, int M, int D, int heads, float ln_eps, float rms_eps,
              void* stream, bool folded = false, bool check = false) {
    const int hd = D / heads;
    const int64_t pD = (int64_t)M * D, p3D = (int64_t)M * 3 * D;
    bool rms_fused = false;
    if (p.qk16) {
        if (!folded) BD_TRY(bd_layernorm(b.x, D, w.ln1_w, w.ln1_b, ln_eps, b.xn, pD, nullptr, 0, M, D, 0, 0, 0, BD_PREC_F16C8, stream));
        {
            bd_gemm_args g = gemm_args(b.xn, D, 0, folded ? w.qkv16_f : w.qkv16, D, 2 * D, b.qkv, 3 * D, 0, 0, M, D, BD_ACT_NONE);      // rows [0, 2D) of the f16 copy
            g.rms_wq = w.q_norm_w; g.rms_wk = w.k_norm_w; g.rms_eps = rms_eps; g.rms_parts = 2;
            rms_fused = hd == 96 && bd_gemm_fuses_qk_rmsnorm(&g, BD_PREC_F16);
            if (!rms_fused) { g.rms_wq = g.rms_wk = nullptr; g.rms_parts = 0; }
            if (folded) ln_consumer(g, b, w.qkv16_s, ln_eps);
            BD_TRY(gemm_or_check(g, BD_PREC_F16, stream, check));
        }
        {
            bd_linear v = folded ? w.qkv_f : w.qkv;                   // rows [2D, 3D) of the F16C8 weight: both planes advance by 2D rows
            v.w = (const unsigned short*)v.w + (int64_t)2 * D * D;
            v.b = v.b + 2 * D;
            bd_gemm_args g = gemm_args(b.xn, D, pD, v, D, D, (unsigned short*)b.qkv + 2 * D, 3 * D, 0, 2 /* f16 plane */, M, D, BD_ACT_NONE);
            g.w_plane = (int64_t)3 * D * D;                           // plane 1 still lies one FULL weight plane behind plane 0
            if (folded) ln_consumer(g, b, w.qkv_s + 2 * D, ln_eps);
            BD_TRY(gemm_or_check(g, BD_PREC_F16C8, stream, check));
        }
    } else {
        if (!folded) BD_TRY(bd_layernorm(b.x, D, w.ln1_w, w.ln1_b, ln_eps, b.xn, pD, nullptr, 0, M, D, 0, 0, 0, p.c_qkv, stream));
        bd_gemm_args g = gemm_args(b.xn, D, pD, folded ? w.qkv_f : w.qkv, D, 3 * D, b.qkv, 3 * D, p3D, p.qkv_out, M, D, BD_ACT_NONE);
        if (w.q_norm_w && hd == 96) {            // q/k RMSNorm in the QKV epilogue where the launch allows it
            g.rms_wq = w.q_norm_w; g.rms_wk = w.k_norm_w; g.rms_eps = rms_eps;
            rms_fused = bd_gemm_fuses_qk_rmsnorm(&g, p.c_qkv);
            if (!rms_fused) g.rms_wq = g.rms_wk = nullptr;
        }
        if (folded) ln_consumer(g, b, w.qkv_s, ln_eps);
        BD_TRY(gemm_or_check(g, p.c_qkv, stream, check));
    }
    if (check) return BD_OK;
    if (w.q_norm_w && !rms_fused) BD_TRY(bd_qk_rmsnorm(b.qkv, p3D, w.q_norm_w, w.k_norm_w, rms_eps, M, heads, hd, p.aprec_in, stream));
    return BD_OK;
}

// may LayerNorm 1 of block `w` be folded (its QKV launches on M rows)?
inline bool ln1_foldable(const bd_block_weights& w, const BlockPlan& p, const BlockBufs& b, int M, int D, int heads, float ln_eps, float rms_eps) {
    if (!w.qkv_f.w || !w.qkv_s || p.c_qkv != BD_PREC_F16C8 || D != 768) return false;
    if (p.qk16 && (!w.qkv16_f.w || !w.qkv16_s)) return false;
    return qkv_stage(w, p, b, M, D, heads, ln_eps, rms_eps, nullptr, true, true) == BD_OK;
}

// The residual side of a block as bd_gemm launches.  3-byte residual stream (bd_gemm_args.ln_resid_in_op, bd_block_weights.ln_resid3): between
// FOLDED LayerNorms nobody reads the stream as fp32 -- the next reader is a Linear that multiplies the operand copy -- so a residual Linear
// may read its residual rows from that copy (b.xn) and write the sum back there only.  What decides, per residual Linear:
//   reads the copy   iff b.xn IS the stream (the LayerNorm in front of it was folded) and the launch has the form (un-promoted F16C8);
//   writes fp32 too  iff someone reads b.x before the next residual Linear: a LayerNorm kernel, a residual Linear without that form, the
//                    stack's final norm / the last decoder block's row gather (need_f32_out).
// x_stale (in / out): b.x does not hold the stream (the previous residual Linear wrote the copy only).
struct ResidPlan { bool fold2, proj_c8, proj_f32, fc2_c8, fc2_f32; };

// split-K scratch: lent to a residual Linear on at most as many rows as the region was carved for (the stream's M; the last decoder
// block's compact rows are fewer)
inline void lend_sk(bd_gemm_args& g, const BlockBufs& b) { if (b.sk && g.M <= BD_SPLITK_MAX_ROWS) g.sk_ws = b.sk; }
inline bd_gemm_args proj_args(const bd_block_weights& w, const BlockBufs& b, float* x, int Mr, int D) {
    bd_gemm_args g = gemm_args(b.ao, D, (int64_t)Mr * D, w.proj, D, D, x, D, 0, 1, Mr, D, BD_ACT_NONE);
    g.resid = x; g.ldr = D;
    lend_sk(g, b);
    return g;
}
inline bd_gemm_args fc2_args(const bd_block_weights& w, const BlockBufs& b, float* x, int Mr, int D) {
    bd_gemm_args g = gemm_args(b.h, 4 * D, (int64_t)Mr * 4 * D, w.fc2, 4 * D, D, x, D, 0, 1, Mr, 4 * D, BD_ACT_NONE);
    g.resid = x; g.ldr = D;
    lend_sk(g, b);
    return g;
}
// the same launch reading its residual from the operand copy (and writing fp32 rows only if asked)
inline void resid_from_copy(bd_gemm_args& g, const BlockBufs& b, int64_t plane, int D, bool f32_too) {
    ln_producer(g, b, plane, D);
    g.ln_resid_in_op = 1; g.resid = nullptr; g.ldr = 0;
    g.out_f32 = f32_too ? 1 : 0;
}
inline bool fold2_possible(const bd_block_weights& w, const BlockPlan& p, const BlockBufs& b, float* x, int Mr, int D, float ln_eps) {
    if (!(w.fc1_f.w && w.fc1_s && p.c_proj == BD_PREC_F16C8 && p.c_fc1 == BD_PREC_F16C8 && p.c_fc2 == BD_PREC_F16C8 && D == 768)) return false;
    bd_gemm_args cp = proj_args(w, b, x, Mr, D), c1 = gemm_args(b.xn, D, (int64_t)Mr * D, w.fc1_f, D, 4 * D, b.h, 4 * D, (int64_t)Mr * 4 * D, 0, Mr, D, BD_ACT_GELU);
    ln_producer(cp, b, (int64_t)Mr * D, D);
    ln_consumer(c1, b, w.fc1_s, ln_eps);
    return bd_gemm_takes_ln_fold(&cp, p.c_proj) && bd_gemm_takes_ln_fold(&c1, p.c_fc1);
}
// can this block's proj read its residual from the operand copy (given that its LayerNorm 1 is folded)?
inline bool proj_c8_possible(const bd_block_weights& w, const BlockPlan& p, const BlockBufs& b, float* x, int Mr, int D, float ln_eps) {
    if (!w.ln_resid3 || !fold2_possible(w, p, b, x, Mr, D, ln_eps)) return false;
    bd_gemm_args g = proj_args(w, b, x, Mr, D);
    resid_from_copy(g, b, (int64_t)Mr * D, D, false);
    return bd_gemm_takes_ln_fold(&g, p.c_proj) != 0;
}
inline ResidPlan plan_resid(const bd_block_weights& w, const BlockPlan& p, const BlockBufs& b, float* x, int Mr, int D, float ln_eps,
                            bool emit_next, bool xn_is_stream, bool next_proj_c8, bool need_f32_out) {
    ResidPlan r{};
    r.fold2 = fold2_possible(w, p, b, x, Mr, D, ln_eps);
    if (r.fold2 && emit_next && w.ln_resid3) {
        bd_gemm_args g = fc2_args(w, b, x, Mr, D);
        resid_from_copy(g, b, (int64_t)Mr * D, D, false);
        r.fc2_c8 = bd_gemm_takes_ln_fold(&g, p.c_fc2) != 0;
    }
    r.proj_c8 = xn_is_stream && proj_c8_possible(w, p, b, x, Mr, D, ln_eps);
    r.proj_f32 = !r.fc2_c8;                                  // (LayerNorm 2 as a kernel implies !fold2, hence !fc2_c8)
    r.fc2_f32 = need_f32_out || !next_proj_c8;
    return r;
}

// x += proj(ao); x += fc2(gelu(fc1(LN2 x)))  on Mr rows (the whole stream, or the query view's compact rows of the last block).
// emit_next: the NEXT block's LayerNorm 1 is folded -- fc2 also writes the operand copy / row statistics of the rows it completes.
// xn_is_stream: b.xn holds the stream's operand copy (this block's LayerNorm 1 was folded); next_proj_c8: the next block's proj can read it
// from there; x_stale: see above (in: b.x is stale, out: it is stale after this block).
int proj_mlp_stage(const bd_block_weights& w, const BlockPlan& p, const BlockBufs& b, float* x, int Mr, int D, float ln_eps, void* stream,
                   bool emit_next = false, bool xn_is_stream = false, bool next_proj_c8 = false, bool need_f32_out = true,
                   bool x_stale = false, bool* x_stale_out = nullptr) {
    const int64_t rD = (int64_t)Mr * D, r4D = (int64_t)Mr * 4 * D;
    const ResidPlan r = plan_resid(w, p, b, x, Mr, D, ln_eps, emit_next, xn_is_stream, next_proj_c8, need_f32_out);
    if (x_stale && !r.proj_c8) return BD_ERR_SHAPE;          // (the previous block's plan looked ahead: cannot happen)
    bd_gemm_args gp = proj_args(w, b, x, Mr, D);
    if (r.proj_c8) resid_from_copy(gp, b, rD, D, r.proj_f32);
    else if (r.fold2) ln_producer(gp, b, rD, D);
    BD_TRY(bd_gemm(&gp, p.c_proj, stream));
    bd_gemm_args g1 = r.fold2 ? gemm_args(b.xn, D, rD, w.fc1_f, D, 4 * D, b.h, 4 * D, r4D, 0, Mr, D, BD_ACT_GELU)
                              : gemm_args(b.xn, D, rD, w.fc1, D, 4 * D, b.h, 4 * D, r4D, handoff_kind(p.c_fc1, p.c_fc2), Mr, D, BD_ACT_GELU);
    if (r.fold2) ln_consumer(g1, b, w.fc1_s, ln_eps);
    else BD_TRY(bd_layernorm(x, D, w.ln2_w, w.ln2_b, ln_eps, b.xn, rD, nullptr, 0, Mr, D, 0, 0, 0, p.c_fc1, stream));
    BD_TRY(bd_gemm(&g1, p.c_fc1, stream));
    bd_gemm_args g2 = fc2_args(w, b, x, Mr, D);
    if (r.fc2_c8) resid_from_copy(g2, b, rD, D, r.fc2_f32);
    else if (emit_next) ln_producer(g2, b, rD, D);
    BD_TRY(bd_gemm(&g2, p.c_fc2, stream));
    if (x_stale_out) *x_stale_out = r.fc2_c8 && !r.fc2_f32;
    return BD_OK;
}

// does the NEXT block's LayerNorm 1 fold behind this block's fc2 (M rows of the stream on both sides)?
inline bool next_ln1_folds(const bd_block_weights& w, const BlockPlan& p, const bd_block_weights* next, int wprec, const BlockBufs& b, int M, int D,
                           int heads, float ln_eps, float rms_eps) {
    if (!next || p.c_fc2 != BD_PREC_F16C8 || D != 768) return false;
    bd_gemm_args g = fc2_args(w, b, b.x, M, D);
    ln_producer(g, b, (int64_t)M * D, D);
    if (!bd_gemm_takes_ln_fold(&g, p.c_fc2)) return false;
    const BlockPlan pn = plan_block(*next, wprec);
    return ln1_foldable(*next, pn, b, M, D, heads, ln_eps, rms_eps);
}

// One pre-LN transformer block: x += proj(attn(LN1 x)); x += fc2(gelu(fc1(LN2 x))).
// BETR: blocks.py:876-886 (+ q/k RMSNorm :257); DINOv2: layers/block.py:89-114 (LayerScale folded).
// n_prefix > 0 (DINOv2: cls + registers lead every image's tokens) and prefix_queries == false: the block's prefix rows are never read
// again (last encoder block), so their attention is skipped (bd_attention_prefix) -- proj and the MLP then run on stale attention
// rows there, whose results nobody consumes (row-wise operators: nothing leaks into the patch rows).
// ln1_folded (in): this block's LayerNorm 1 is folded (the previous block's fc2 emitted for it); returns through *next_folded whether the
// next block's is (this block's fc2 then emitted).
// x_stale (in / out) and next_needs_f32: the 3-byte residual stream (proj_mlp_stage); next_compact: the next block runs its residual side on
// other rows (the last decoder block), so its proj cannot read this block's copy.
int run_block(const bd_block_weights& w, const BlockBufs& b, int M, int batch, int seq, int D, int heads, float ln_eps, float rms_eps,
              int wprec, void* stream, int n_prefix = 0, bool prefix_queries = true, bool ln1_folded = false,
              const bd_block_weights* next = nullptr, bool* next_folded = nullptr, bool* x_stale = nullptr, bool next_compact = false,
              bool latency = false) {
    const BlockPlan p = plan_block(w, wprec);
    const int hd = D / heads;
    BD_TRY(qkv_stage(w, p, b, M, D, heads, ln_eps, rms_eps, stream, ln1_folded));
    const float scale = 1.0f / sqrtf((float)hd);
    if (n_prefix > 0 && seq > n_prefix && !prefix_queries)
        BD_TRY(bd_attention_prefix(b.qkv, (int64_t)M * 3 * D, b.ao, (int64_t)M * D, batch, seq, heads, hd, scale, n_prefix, 0, p.aprec, stream));
    else
        BD_TRY(bd_attention_q_forms(b.qkv, (int64_t)M * 3 * D, b.ao, (int64_t)M * D, batch, seq, heads, hd, scale, nullptr, seq, p.aprec,
                                    latency ? 1 : 0, stream));
    const bool emit = next_ln1_folds(w, p, next, wprec, b, M, D, heads, ln_eps, rms_eps);
    if (next_folded) *next_folded = emit;
    bool next_c8 = false;
    if (emit && next && !next_compact) {
        const BlockPlan pn = plan_block(*next, wprec);
        next_c8 = proj_c8_possible(*next, pn, b, b.x, M, D, ln_eps);
    }
    const bool stale_in = x_stale ? *x_stale : false;
    // fp32 rows after this block: the stack's last block (final norm / head), or a next block that gathers rows from b.x
    return proj_mlp_stage(w, p, b, b.x, M, D, ln_eps, stream, emit, ln1_folded, next_c8, /*need_f32_out=*/next == nullptr || next_compact, stale_in, x_stale);
}

// Last decoder block: its output is consumed for the query view only (betr.py:303), so only K/V need every token.
// LN1 + QKV (+ q/k RMSNorm) run on all rows; attention takes queries from the query view's P rows and writes a
// compact [B*P, D] result; proj, LN2 and the MLP then run on B*P rows (1/T of the work).  Row-wise arithmetic is
// unchanged, so the result is bit-identical to the full-width block.  xc: fp32 [B*P, D] compact residual stream.
int run_last_block_query_only(const bd_block_weights& w, const BlockBufs& b, float* xc, const int32_t* query_idx, int B, int T, int P,
                              int D, int heads, float ln_eps, float rms_eps, int wprec, void* stream, bool ln1_folded = false,
                              bool latency = false) {
    const BlockPlan p = plan_block(w, wprec);
    const int hd = D / heads, M = B * T * P, Mq = B * P;
    BD_TRY(qkv_stage(w, p, b, M, D, heads, ln_eps, rms_eps, stream, ln1_folded));
    BD_TRY(bd_attention_q_forms(b.qkv, (int64_t)M * 3 * D, b.ao, (int64_t)Mq * D, B, T * P, heads, hd, 1.0f / sqrtf((float)hd), query_idx, P,
                                p.aprec, latency ? 1 : 0, stream));
    BD_TRY(bd_gather_query_rows_f32(b.x, query_idx, xc, B, T, P, D, stream));
    return proj_mlp_stage(w, p, b, xc, Mq, D, ln_eps, stream);
}

BlockBufs carve_block(Carver& c, int64_t M, int D, int np, bool latency) {
    BlockBufs b;
    b.x = (float*)c.take((size_t)M * D * 4);
    b.xn = c.take((size_t)M * D * 2 * np);
    b.qkv = c.take((size_t)M * 3 * D * 2 * np);
    b.ao = c.take((size_t)M * D * 2 * np);
    b.h = c.take((size_t)M * 4 * D * 2 * np);
    b.st = (float*)c.take((size_t)M * ((D + 95) / 96) * 2 * 4);
    // split-K scratch for the stream's residual Linears when the caller opted into the latency forms (bd_*_weights.latency_mode) and the
    // whole stream is a few tiles (one or two poses at a time)
    const size_t skb = (latency && np == 2 && M <= BD_SPLITK_MAX_ROWS) ? bd_gemm_splitk_workspace_bytes((int)M, D) : 0;
    b.sk = skb ? c.take(skb) : nullptr;
    b.sk_flag_bytes = skb ? bd_gemm_splitk_flag_bytes((int)M, D) : 0;
    return b;
}

struct EncBufs { void* a_patch; BlockBufs blk; size_t bytes; };
EncBufs carve_encoder(const bd_dino_weights* w, int n, int prec, void* ws) {
    Carver c{(unsigned char*)ws, 0};
    const int np = planes_of(prec), P = w->grid * w->grid;
    EncBufs e;
    e.a_patch = c.take((size_t)n * P * w->kpad * 2 * np);
    e.blk = carve_block(c, (int64_t)n * (P + w->n_prefix), w->dim, np, w->latency_mode != 0);
    e.bytes = c.off + 256;
    return e;
}

struct DecBufs { void *a_heat, *t1, *qtok; float *t2, *rgb, *proj; BlockBufs blk; size_t bytes; };
DecBufs carve_decoder(const bd_betr_weights* w, int B, int T, int prec, void* ws) {
    Carver c{(unsigned char*)ws, 0};
    const int np = planes_of(prec), P = w->grid * w->grid, D = w->dim;
    const int64_t Mb = (int64_t)B * T * P, Mq = (int64_t)B * P;
    const int F = w->patch * w->patch * w->box_dim;
    DecBufs d;
    d.a_heat = c.take((size_t)Mb * w->kpad * 2 * np);
    d.t1 = c.take((size_t)Mb * D * 2 * np);
    d.t2 = (float*)c.take((size_t)Mb * D * 4);
    d.rgb = (float*)c.take((size_t)Mb * D * 4);
    d.qtok = c.take((size_t)Mq * D * 2 * np);
    d.proj = (float*)c.take((size_t)Mq * F * 4);
    d.blk = carve_block(c, Mb, D, np, w->latency_mode != 0);
    d.bytes = c.off + 256;
    return d;
}

inline bool bad_prec(int prec) {
    return prec != BD_PREC_BF16 && prec != BD_PREC_F16 && prec != BD_PREC_BF16X3 && prec != BD_PREC_FP8 &&
           prec != BD_PREC_BF16X3_ATTN_X3 && prec != BD_PREC_F16C8 && prec != BD_PREC_F16C8_QK16 && prec != BD_PREC_F16X3 &&
           prec != BD_PREC_F16X3_ATTN_X3;
}

}  // namespace

extern "C" int bd_abi_version(void) { return BD_ABI_VERSION; }
extern "C" const char* bd_target_arch(void) { return "gfx950"; }

extern "C" size_t bd_encoder_workspace_bytes(const bd_dino_weights* w, int n_images, int prec) {
    if (!w || n_images <= 0 || bad_prec(prec)) return 0;
    return carve_encoder(w, n_images, prec, nullptr).bytes;
}

extern "C" int bd_encoder_forward(const bd_dino_weights* w, const void* images, int img_dtype, int n_images,
                                  int size, float* feats32, void* feats16, int64_t feats16_plane, void* workspace,
                                  size_t workspace_bytes, int wprec, void* stream) {
    if (!w || !images || !workspace || !w->blocks || (!feats32 && !feats16)) return BD_ERR_NULL;
    if (bad_prec(wprec)) return BD_ERR_DTYPE;
    const int prec = gemm_prec(wprec);      // operand class of the unit operators; wprec also carries the attention policy
    if (n_images <= 0 || size != w->grid * w->patch || w->dim % w->heads || w->kpad % 64 ||
        w->kpad < 3 * w->patch * w->patch)
        return BD_ERR_SHAPE;
    if (w->feats_prec != 0 && !(w->feats_prec == prec || w->feats_prec == promoted_class(prec))) return BD_ERR_DTYPE;
    const int feats_prec = w->feats_prec ? w->feats_prec : prec;
    if ((uintptr_t)workspace & 255) return BD_ERR_ALIGN;
    const EncBufs e = carve_encoder(w, n_images, prec, workspace);
    if (workspace_bytes < e.bytes) return BD_ERR_WORKSPACE;
    const int P = w->grid * w->grid, D = w->dim, tpi = P + w->n_prefix;
    const int Mp = n_images * P, Md = n_images * tpi;
    if (e.blk.sk && hipMemsetAsync(e.blk.sk, 0, e.blk.sk_flag_bytes, (hipStream_t)stream) != hipSuccess) return BD_ERR_WORKSPACE;

    // K1+K2: normalise + im2col, then the patch-embed GEMM scattering rows b*P+p -> b*tpi+n_prefix+p and
    // adding the (pre-resampled) positional table  (vision_transformer.py:213-232, patch_embed.py:65-75)
    const int c_pe = lin_class(prec, (prec == BD_PREC_F16C8 || prec == BD_PREC_FP8) ? w->promote_misc : 0, BD_PROMOTE_PATCH_EMBED);
    BD_TRY(bd_im2col_images(images, img_dtype, e.a_patch, (int64_t)Mp * w->kpad, n_images, size, w->patch, w->kpad,
                            c_pe, stream));
    {
        bd_gemm_args g = gemm_args(e.a_patch, w->kpad, (int64_t)Mp * w->kpad, w->patch_embed, w->kpad, D, e.blk.x, D, 0,
                                   1, Mp, w->kpad, BD_ACT_NONE);
        g.addtab = w->pos_patch; g.tab_rows = P;
        g.rpg_in = P; g.rpg_out = tpi; g.row_off = w->n_prefix;
        BD_TRY(bd_gemm(&g, c_pe, stream));
    }
    BD_TRY(bd_write_prefix_tokens(e.blk.x, w->prefix_tokens, n_images, tpi, w->n_prefix, D, stream));
    bool folded = false, stale = false;      // LayerNorm 1 of block i is folded behind block i-1's fc2 (never block 0's); b.x is stale
    for (int i = 0; i < w->depth; ++i) {
        bool next_folded = false;
        BD_TRY(run_block(w->blocks[i], e.blk, Md, n_images, tpi, D, w->heads, w->ln_eps, 0.f, wprec, stream, w->n_prefix, i + 1 < w->depth,
                         folded, i + 1 < w->depth ? &w->blocks[i + 1] : nullptr, &next_folded, &stale));
        folded = next_folded;
    }
    if (stale) return BD_ERR_SHAPE;          // (the last block writes fp32 rows: the final norm reads them)
    // final LayerNorm on the patch tokens only (vision_transformer.py:263-267); feats16 in the class the consumer's first Linear reads
    BD_TRY(bd_layernorm(e.blk.x, D, w->norm_w, w->norm_b, w->ln_eps, feats16, feats16_plane, feats32, D, Mp, D, P, tpi,
                        w->n_prefix, feats_prec, stream));
    return BD_OK;
}

extern "C" size_t bd_decoder_workspace_bytes(const bd_betr_weights* w, int B, int T, int prec) {
    if (!w || B <= 0 || T <= 0 || bad_prec(prec)) return 0;
    return carve_decoder(w, B, T, prec, nullptr).bytes;
}

extern "C" int bd_decoder_forward(const bd_betr_weights* w, const void* bbox_feat, int in_dtype, const void* feats16,
                                  int64_t feats16_plane, const int32_t* query_idx, int B, int T, int size,
                                  float* logits, float* heat, void* workspace, size_t workspace_bytes, int wprec,
                                  void* stream) {
    if (!w || !bbox_feat || !feats16 || !query_idx || !workspace || !w->blocks || (!logits && !heat)) return BD_ERR_NULL;
    if (bad_prec(wprec)) return BD_ERR_DTYPE;
    const int prec = gemm_prec(wprec);
    if (B <= 0 || T <= 0 || size != w->grid * w->patch || w->dim % w->heads || w->kpad % 64 || w->box_dim != 8 ||
        w->kpad < w->patch * w->patch * w->box_dim)
        return BD_ERR_SHAPE;
    if ((uintptr_t)workspace & 255) return BD_ERR_ALIGN;
    const DecBufs d = carve_decoder(w, B, T, prec, workspace);
    if (workspace_bytes < d.bytes) return BD_ERR_WORKSPACE;
    const int P = w->grid * w->grid, D = w->dim, F = w->patch * w->patch * w->box_dim;
    const int Mb = B * T * P, Mq = B * P;
    const int64_t pD = (int64_t)Mb * D;
    if (d.blk.sk && hipMemsetAsync(d.blk.sk, 0, d.blk.sk_flag_bytes, (hipStream_t)stream) != hipSuccess) return BD_ERR_WORKSPACE;

    // operand classes of the Linears outside the blocks (F16C8 family: per-Linear promotion, include/boxdreamer_hip.h)
    const int pm0 = (prec == BD_PREC_F16C8 || prec == BD_PREC_FP8) ? w->promote_misc : 0;
    const int pmisc = (pm0 & BD_PROMOTE_ADAPTER_FC1) ? (pm0 | BD_PROMOTE_ADAPTER_FC2) : pm0;
    const int c_a1 = lin_class(prec, pmisc, BD_PROMOTE_ADAPTER_FC1), c_a2 = lin_class(prec, pmisc, BD_PROMOTE_ADAPTER_FC2);
    const int c_be = lin_class(prec, pmisc, BD_PROMOTE_BBOX_EMB), c_bp = lin_class(prec, pmisc, BD_PROMOTE_BBOX_PROJ);
    // K6 adapter: LN_noaffine(fc2(gelu(fc1(feat))))  (betr.py:313-317); feats16 arrives in the class of adapter fc1
    {
        bd_gemm_args g = gemm_args(feats16, D, feats16_plane, w->adapter_fc1, D, D, d.t1, D, pD, handoff_kind(c_a1, c_a2), Mb, D, BD_ACT_GELU);
        BD_TRY(bd_gemm(&g, c_a1, stream));
    }
    {
        bd_gemm_args g = gemm_args(d.t1, D, pD, w->adapter_fc2, D, D, d.t2, D, 0, 1, Mb, D, BD_ACT_NONE);
        BD_TRY(bd_gemm(&g, c_a2, stream));
    }
    BD_TRY(bd_layernorm(d.t2, D, nullptr, nullptr, w->adapter_ln_eps, nullptr, 0, d.rgb, D, Mb, D, 0, 0, 0, prec, stream));
    // K7+K8: heatmap patch embedding fused with  + rgb + pos  (betr.py:324-329, 367-399)
    BD_TRY(bd_patchify_heatmaps(bbox_feat, in_dtype, d.a_heat, (int64_t)Mb * w->kpad, B * T, w->box_dim, size, w->patch,
                                w->kpad, c_be, stream));
    {
        bd_gemm_args g = gemm_args(d.a_heat, w->kpad, (int64_t)Mb * w->kpad, w->bbox_emb, w->kpad, D, d.blk.x, D, 0, 1,
                                   Mb, w->kpad, BD_ACT_NONE);
        g.addtab = w->pos_table; g.tab_rows = P;
        g.resid = d.rgb; g.ldr = D;
        BD_TRY(bd_gemm(&g, c_be, stream));
    }
    BD_TRY(bd_query_substitute(d.blk.x, d.rgb, w->pos_table, w->query_token, query_idx, B, T, P, D, stream));
    // K9: joint self-attention over all T*P tokens of a sample
    // (latency forms, one or two poses per call: the q, k / v column split of BD_PREC_F16C8_QK16 is two launches of ~20 us each where one
    // F16C8 QKV launch takes ~26 us -- at these sizes a launch costs its fixed part, not its passes; q, k then carry the full F16C8 product)
    const bool lat = w->latency_mode && Mb <= BD_SPLITK_MAX_ROWS;
    const int bprec = (lat && wprec == BD_PREC_F16C8_QK16) ? BD_PREC_F16C8 : wprec;
    bool folded = false, stale = false;      // LayerNorm 1 of block i is folded behind block i-1's fc2 (never block 0's); b.x is stale
    for (int i = 0; i + 1 < w->depth; ++i) {
        bool next_folded = false;
        BD_TRY(run_block(w->blocks[i], d.blk, Mb, B, T * P, D, w->heads, w->ln_eps, w->rms_eps, bprec, stream, 0, true, folded,
                         &w->blocks[i + 1], &next_folded, &stale, /*next_compact=*/i + 2 == w->depth, lat));
        folded = next_folded;
    }
    if (stale) return BD_ERR_SHAPE;          // (the block in front of the last one writes fp32 rows: the last block gathers its query rows from them)
    // last block: query-view rows only past the K/V projection; d.t2 (dead since the adapter) holds the compact stream
    BD_TRY(run_last_block_query_only(w->blocks[w->depth - 1], d.blk, d.t2, query_idx, B, T, P, D, w->heads,
                                     w->ln_eps, w->rms_eps, bprec, stream, folded, lat));
    // K10: head on the query view's tokens (no final norm, betr.py:298-306)
    BD_TRY(bd_gather_query_tokens(d.t2, nullptr, d.qtok, (int64_t)Mq * D, B, 1, P, D, c_bp, stream));
    {
        bd_gemm_args g = gemm_args(d.qtok, D, (int64_t)Mq * D, w->bbox_proj, D, F, d.proj, F, 0, 1, Mq, D, BD_ACT_NONE);
        BD_TRY(bd_gemm(&g, c_bp, stream));
    }
    BD_TRY(bd_unpatchify_sigmoid(d.proj, logits, heat, B, w->box_dim, size, w->patch, stream));
    return BD_OK;
}

// ----------------------------------------------------------------------------------------------------------------------------
// Sub-batch lanes.  Samples are independent all the way down the path, and every row's result is independent
// of the tile shape its launch picks (tests/test_gpu_ops.py::test_gemm_row_result_independent_of_tile_shape, the batch-invariance
// tests), so ONE batch may run as `lanes` contiguous sub-batches on `lanes` streams without changing a bit of the result: lane 0 on
// the caller's stream, the others on side streams this library owns, forked from and joined back into the caller's stream with
// events inside the call -- the call stays stream-ordered for the caller (and capturable: an event wait pulls the side stream into
// the caller's capture).  What it buys: the kernels of one lane run on the CUs the other lane's ragged last round leaves idle and
// the HBM-bound launches (LayerNorm) of one lane overlap the MFMA-bound launches of the other -- what two BATCHES in flight bought
// in rounds 2-4, now inside one batch of configs[1] (profiles/r4_subbatch_lanes.md).
namespace {

constexpr int kMaxLanes = 4, kMaxDevices = 64;
struct LaneSet {
    hipStream_t side[kMaxLanes - 1];
    hipEvent_t fork, join[kMaxLanes - 1];
    bool ready = false;
};
// Per HOST THREAD and device (round 6; process-global until then): a thread that captures `stream` into a HIP graph pulls ITS OWN side
// streams into the capture and nobody else's -- two threads may enqueue laned calls on one device at the same time, capturing or not.
// Created on first use (or by bd_lanes_prepare), never destroyed: a thread that exits leaves its three streams / four events behind.
thread_local LaneSet g_lanes[kMaxDevices];

int lanes_of_current_device(LaneSet** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return BD_ERR_SHAPE;
    LaneSet& L = g_lanes[dev];
    if (!L.ready) {
        if ((e = hipEventCreateWithFlags(&L.fork, hipEventDisableTiming)) != hipSuccess) return (int)e;
        for (int i = 0; i < kMaxLanes - 1; ++i) {
            if ((e = hipStreamCreateWithFlags(&L.side[i], hipStreamNonBlocking)) != hipSuccess) return (int)e;
            if ((e = hipEventCreateWithFlags(&L.join[i], hipEventDisableTiming)) != hipSuccess) return (int)e;
        }
        L.ready = true;
    }
    *out = &L;
    return BD_OK;
}

inline int lane_count(int lanes, int units) { return lanes < 1 ? 1 : (lanes > kMaxLanes ? kMaxLanes : (lanes > units ? units : lanes)); }
inline int lane_units(int units, int lanes, int l) { return units / lanes + (l < units % lanes ? 1 : 0); }
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// element offset e0 into a 16-bit operand buffer of class `cls`: plane 0 moves by e0 elements; F16C8's one-byte lo8 plane (which
// starts `plane` 2-byte units behind plane 0) moves by e0 BYTES, i.e. the plane distance seen from the moved base shrinks by e0 / 2
inline void operand_slice(const void* base, int64_t plane, int cls, int64_t e0, const void** b, int64_t* p) {
    const int esz = cls == BD_PREC_FP8 ? 1 : 2;
    *b = base ? (const unsigned char*)base + e0 * esz : nullptr;
    *p = cls == BD_PREC_F16C8 ? plane - e0 / 2 : plane;
}
inline int dtype_bytes(int dt) { return dt == BD_DTYPE_F32 ? 4 : 2; }

// fork / join around the per-lane calls.  Every lane is enqueued even after an earlier one failed, and the joins are always
// recorded, so that a capture in progress is left well-formed; the first error is returned.
template <class F> int run_lanes(int lanes, hipStream_t main, F&& lane_call) {
    LaneSet* L = nullptr;
    BD_TRY(lanes_of_current_device(&L));
    hipError_t e = hipEventRecord(L->fork, main);
    if (e != hipSuccess) return (int)e;
    int rc = BD_OK;
    bd_concurrent_launches() = lanes;          // the lanes' launches run side by side: tile-form choices count a lane's share of the CUs
    for (int l = 0; l < lanes; ++l) {
        hipStream_t s = l == 0 ? main : L->side[l - 1];
        if (l > 0 && (e = hipStreamWaitEvent(s, L->fork, 0)) != hipSuccess && rc == BD_OK) rc = (int)e;
        const int r = lane_call(l, s);
        if (r != BD_OK && rc == BD_OK) rc = r;
    }
    bd_concurrent_launches() = 1;
    for (int l = 1; l < lanes; ++l) {
        if ((e = hipEventRecord(L->join[l - 1], L->side[l - 1])) != hipSuccess && rc == BD_OK) rc = (int)e;
        if ((e = hipStreamWaitEvent(main, L->join[l - 1], 0)) != hipSuccess && rc == BD_OK) rc = (int)e;
    }
    return rc;
}

}  // namespace

extern "C" int bd_lanes_prepare(void) {
    LaneSet* L = nullptr;
    return lanes_of_current_device(&L);
}

extern "C" size_t bd_encoder_workspace_bytes_lanes(const bd_dino_weights* w, int n_images, int prec, int lanes) {
    if (!w || n_images <= 0 || bad_prec(prec)) return 0;
    const int nl = lane_count(lanes, n_images);
    size_t total = 0;
    for (int l = 0; l < nl; ++l) total += align256(carve_encoder(w, lane_units(n_images, nl, l), gemm_prec(prec), nullptr).bytes);
    return total;
}

extern "C" int bd_encoder_forward_lanes(const bd_dino_weights* w, const void* images, int img_dtype, int n_images, int size,
                                        float* feats32, void* feats16, int64_t feats16_plane, void* workspace,
                                        size_t workspace_bytes, int wprec, int lanes, void* stream) {
    const int nl = lane_count(lanes, n_images);
    if (nl <= 1 || !w) return bd_encoder_forward(w, images, img_dtype, n_images, size, feats32, feats16, feats16_plane, workspace,
                                                 workspace_bytes, wprec, stream);
    if (bad_prec(wprec)) return BD_ERR_DTYPE;
    if (!images || !workspace) return BD_ERR_NULL;
    if (img_dtype < 0 || img_dtype > 2) return BD_ERR_DTYPE;
    if ((uintptr_t)workspace & 255) return BD_ERR_ALIGN;
    if (workspace_bytes < bd_encoder_workspace_bytes_lanes(w, n_images, wprec, nl)) return BD_ERR_WORKSPACE;
    const int prec = gemm_prec(wprec), fcls = w->feats_prec ? w->feats_prec : prec;
    const int64_t PD = (int64_t)w->grid * w->grid * w->dim, img_elems = (int64_t)3 * size * size;
    return run_lanes(nl, (hipStream_t)stream, [&](int l, hipStream_t s) {
        int first = 0;
        size_t woff = 0;
        for (int j = 0; j < l; ++j) {
            first += lane_units(n_images, nl, j);
            woff += align256(carve_encoder(w, lane_units(n_images, nl, j), prec, nullptr).bytes);
        }
        const int n = lane_units(n_images, nl, l);
        const void* f16 = nullptr;
        int64_t plane = feats16_plane;
        operand_slice(feats16, feats16_plane, fcls, first * PD, &f16, &plane);
        return bd_encoder_forward(w, (const unsigned char*)images + first * img_elems * dtype_bytes(img_dtype), img_dtype, n, size,
                                  feats32 ? feats32 + first * PD : nullptr, const_cast<void*>(f16), plane,
                                  (unsigned char*)workspace + woff, align256(carve_encoder(w, n, prec, nullptr).bytes), wprec, s);
    });
}

extern "C" size_t bd_decoder_workspace_bytes_lanes(const bd_betr_weights* w, int B, int T, int prec, int lanes) {
    if (!w || B <= 0 || T <= 0 || bad_prec(prec)) return 0;
    const int nl = lane_count(lanes, B);
    size_t total = 0;
    for (int l = 0; l < nl; ++l) total += align256(carve_decoder(w, lane_units(B, nl, l), T, gemm_prec(prec), nullptr).bytes);
    return total;
}

extern "C" int bd_decoder_forward_lanes(const bd_betr_weights* w, const void* bbox_feat, int in_dtype, const void* feats16,
                                        int64_t feats16_plane, const int32_t* query_idx, int B, int T, int size, float* logits,
                                        float* heat, void* workspace, size_t workspace_bytes, int wprec, int lanes, void* stream) {
    const int nl = lane_count(lanes, B);
    if (nl <= 1 || !w) return bd_decoder_forward(w, bbox_feat, in_dtype, feats16, feats16_plane, query_idx, B, T, size, logits, heat,
                                                 workspace, workspace_bytes, wprec, stream);
    if (bad_prec(wprec)) return BD_ERR_DTYPE;
    if (!bbox_feat || !feats16 || !query_idx || !workspace) return BD_ERR_NULL;
    if (in_dtype < 0 || in_dtype > 2) return BD_ERR_DTYPE;
    if (T <= 0) return BD_ERR_SHAPE;
    if ((uintptr_t)workspace & 255) return BD_ERR_ALIGN;
    if (workspace_bytes < bd_decoder_workspace_bytes_lanes(w, B, T, wprec, nl)) return BD_ERR_WORKSPACE;
    const int prec = gemm_prec(wprec);
    const int pm0 = (prec == BD_PREC_F16C8 || prec == BD_PREC_FP8) ? w->promote_misc : 0;
    const int fcls = lin_class(prec, pm0, BD_PROMOTE_ADAPTER_FC1);       // the class bd_decoder_forward reads feats16 in
    const int64_t PD = (int64_t)w->grid * w->grid * w->dim, view_elems = (int64_t)w->box_dim * size * size;
    return run_lanes(nl, (hipStream_t)stream, [&](int l, hipStream_t s) {
        int first = 0;
        size_t woff = 0;
        for (int j = 0; j < l; ++j) {
            first += lane_units(B, nl, j);
            woff += align256(carve_decoder(w, lane_units(B, nl, j), T, prec, nullptr).bytes);
        }
        const int b = lane_units(B, nl, l);
        const void* f16 = nullptr;
        int64_t plane = feats16_plane;
        operand_slice(feats16, feats16_plane, fcls, (int64_t)first * T * PD, &f16, &plane);
        return bd_decoder_forward(w, (const unsigned char*)bbox_feat + (int64_t)first * T * view_elems * dtype_bytes(in_dtype), in_dtype,
                                  f16, plane, query_idx + first, b, T, size, logits ? logits + first * view_elems : nullptr,
                                  heat ? heat + first * view_elems : nullptr, (unsigned char*)workspace + woff,
                                  align256(carve_decoder(w, b, T, prec, nullptr).bytes), wprec, s);
    });
}
