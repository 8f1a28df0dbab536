"""GPU parity of the whole corner-heatmap path (encoder -> decoder -> decode) through the reference
plugin surface, against the CPU oracle on the same seeded weights/inputs and against the committed
golden fixtures (generated from the real reference by oracle/make_golden.py).

Tolerances (max-abs on the heatmap LOGITS, written here as the contract):
  f16c8 (strict mode)   <= 1e-3   -- north_star's bar (f16 pass + e4m3 correction pass; measured ~2e-4 at full depth)
  bf16x3 (round-1 strict) <= 1e-3 -- split-bf16, 3 passes (measured ~1.3e-4)
  fp16                  <= 2.5e-2 -- one f16 MFMA pass (operand rounding 2^-11)
  bf16                  <= 1e-1   -- one bf16 MFMA pass (operand rounding 2^-8); the reference's own
                                     bf16-autocast forward is 2.2e-2 off its fp32 forward on the +-1
                                     heatmap
"""
import json
import os

import numpy as np
import pytest
import torch

from boxdreamer_amd import hip_ops, synth
from boxdreamer_amd.betr import BETR
from boxdreamer_amd.encoder import DinoV2Wrapper
from oracle import boxdreamer_oracle as orc

pytestmark = pytest.mark.gpu
LOGIT_TOL = {"f16x3": 1e-3, "f16x3_attn_x3": 1e-3, "f16c8_qk16": 1e-3, "f16c8": 1e-3, "bf16x3": 1e-3, "bf16x3_attn_x3": 1e-3, "fp16": 2.5e-2, "bf16": 1e-1}
FEAT_TOL = {"f16x3": 1e-3, "f16x3_attn_x3": 1e-3, "f16c8_qk16": 1e-3, "f16c8": 1e-3, "bf16x3": 1e-3, "bf16x3_attn_x3": 1e-3, "fp16": 2.5e-2, "bf16": 1e-1}
STRICT = ("f16x3", "f16x3_attn_x3", "f16c8_qk16", "f16c8", "bf16x3", "bf16x3_attn_x3")
REPORT = {}


def assert_identical_topk_sets(idx, o, err, tag):
    """north_star: "identical argmax corner indices" (the reference decodes the top-20 SET of every corner map, box_utils.py:87-95).
    Every map's set must equal the oracle's.  The one computed exemption: a map where the ORACLE's own 20th and 21st logits are
    closer than 2 x the measured logit error of this run -- there the reference's choice is itself decided below the error any
    finite-precision forward carries (random-weight heatmaps are noise: such near-ties exist).  The gap is printed.  Returns the
    fraction of identical maps."""
    eq = (idx.sort(-1)[0] == o["topk_idx"].sort(-1)[0]).all(-1)            # (B, 8)
    for b, c in (~eq).nonzero().tolist():
        top = o["logits"][b, c].flatten().topk(21)[0]
        gap = (top[19] - top[20]).item()
        print(f"[{tag}] sample {b} corner {c}: top-20 set differs; the oracle's 20th / 21st logits are {gap:.3e} apart (this run's logit error: {err:.3e})")
        assert gap <= 2.0 * err, (tag, b, c, gap, err)
    return eq.float().mean().item()


def _build(prec, dino_depth, betr_depth):
    enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "synthetic_seed": 4321, "depth": dino_depth,
                               "hip_precision": prec})
    dec = BETR(d_model=768, nhead=8, num_decoder_layers=betr_depth, decoder_only=True, patch_size=14, img_size=224,
               diff_emb=False, nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True,
               patchify_rays=True, pose_representation="bb8", bbox_representation="heatmap", hip_precision=prec)
    dec.load_state_dict(synth.betr_state_dict(seed=1234, depth=betr_depth), strict=True)
    return enc, dec.cuda().eval()


def _run(prec, B, T, dino_depth, betr_depth, seed, in_dtype=torch.float32):
    enc, dec = _build(prec, dino_depth, betr_depth)
    data = synth.make_batch(seed=seed, B=B, T=T)
    mask = torch.zeros(B, T, dtype=torch.bool)
    mask[torch.arange(B), data["query_idx"]] = True
    img, bf = data["images"].to(in_dtype).cuda(), data["bbox_feat"].to(in_dtype).cuda()
    feats = enc.predict(img)
    heat = dec(bf, img, mask.cuda(), feats, None)
    kp, kn, idx = hip_ops.decode_topk(heat)
    torch.cuda.synchronize()
    return data, feats.cpu(), dec.last_logits.cpu(), heat.cpu(), kp.cpu(), kn.cpu(), idx.cpu().long()


_ORACLE = {}     # the fp32 CPU oracle's outputs per distinct (depths, inputs): every precision mode of a case is compared with the same ones


def _oracle(data, dino_depth, betr_depth):
    key = (dino_depth, betr_depth, tuple(data["images"].shape), float(data["images"].double().sum()),
           float(data["bbox_feat"].double().sum()), tuple(int(q) for q in data["query_idx"].flatten()))
    if key not in _ORACLE:
        _ORACLE[key] = orc.boxdreamer_forward(data, synth.betr_state_dict(1234, betr_depth), synth.dino_state_dict(4321, dino_depth))
    return _ORACLE[key]


@pytest.mark.parametrize("prec", ["f16c8_qk16", "f16c8", "f16x3", "f16x3_attn_x3", "bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("case", ["tiny_d2_T2", "tiny_d2_T3_B2", "full_T2", "full_T6"])
def test_path_vs_oracle_and_golden(hip, golden_dir, prec, case):
    g = np.load(os.path.join(golden_dir, f"case_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    B, T, dd, bd, seed = meta["B"], meta["T"], meta["dino_depth"], meta["betr_depth"], meta["input_seed"]
    data, feats, logits, heat, kp, kn, idx = _run(prec, B, T, dd, bd, seed)
    o = _oracle(data, dd, bd)
    e_feat = (feats - o["rgb_feat"]).abs().max().item()
    e_logit = (logits - o["logits"]).abs().max().item()
    e_heat = (heat - o["heat"]).abs().max().item()
    same = (idx.sort(-1)[0] == o["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
    e_kp = (kp - o["corners_px"]).abs().max().item()
    # against the REFERENCE's own outputs (fixtures), not just the restatement
    e_gold = np.abs(logits.reshape(B, -1)[:, ::7].numpy() - g["logits_strided"]).max()
    e_gold_feat = np.abs(feats.reshape(B, -1)[:, ::97].numpy() - g["rgb_feat_strided"]).max()
    REPORT[f"{case}/{prec}"] = dict(feat=e_feat, logits=e_logit, heat=e_heat, top20_sets_equal=same, kp_px=e_kp,
                                    logits_vs_golden=float(e_gold), feat_vs_golden=float(e_gold_feat))
    print(f"[{case} {prec}] " + json.dumps(REPORT[f"{case}/{prec}"]))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.json", "w") as f:
        json.dump(REPORT, f, indent=1)
    assert e_feat <= FEAT_TOL[prec], e_feat
    assert e_logit <= LOGIT_TOL[prec], e_logit
    assert e_gold <= LOGIT_TOL[prec] + 1e-4, e_gold
    assert e_gold_feat <= FEAT_TOL[prec] + 1e-4
    if prec in STRICT:
        # identical top-20 index sets (computed exemption for oracle near-ties only); a swapped index moves a corner by <= 224/20 px
        assert_identical_topk_sets(idx, o, e_logit, f"{case} {prec}")
        assert e_kp <= 224 / 20 * 2, (same, e_kp)
        gk = np.abs(kp.numpy() - g["corners_px"]).max()
        assert gk <= 224 / 20 * 2


@pytest.mark.parametrize("prec,tol", [("bf16x3_attn_x3", 1e-3)])
def test_strict_attention_policies_full_T6(hip, golden_dir, prec, tol):
    """The attention policies of the strict family are `prec` values of the whole-path entry points (they used to be an
    environment switch inside the library): split-bf16 attention everywhere meets the 1e-3 bar with the largest margin.
    (f16 attention everywhere -- ABI <= 6's bf16x3_attn_f16 -- measured ~1.2e-3 and was removed in ABI 7.)"""
    g = np.load(os.path.join(golden_dir, "case_full_T6.npz"))
    meta = json.loads(str(g["meta"]))
    data, feats, logits, heat, kp, kn, idx = _run(prec, meta["B"], meta["T"], meta["dino_depth"], meta["betr_depth"],
                                                  meta["input_seed"])
    e_gold = float(np.abs(logits.reshape(meta["B"], -1)[:, ::7].numpy() - g["logits_strided"]).max())
    REPORT[f"full_T6/{prec}"] = dict(logits_vs_golden=e_gold)
    print(f"[full_T6 {prec}] logits vs the reference's fixture: {e_gold:.3e}")
    assert e_gold <= tol, e_gold


@pytest.mark.parametrize("prec", ["f16c8_qk16", "bf16x3"])
def test_views17_reduced_depth(hip, prec):
    """BASELINE configs[3] shape: 1 query + 16 references (T = 17, one BETR sequence of 4352 tokens = 68 key tiles per
    attention row block) at reduced depth (2 + 2 layers, so the CPU oracle finishes in seconds), in the default (strict) mode and
    round 1's, B = 2 with the query view in the middle of the list for one sample."""
    dd, bd, B, T = 2, 2, 2, 17
    enc, dec = _build(prec, dd, bd)
    data = synth.make_batch(seed=51, B=B, T=T)
    data["query_idx"] = torch.tensor([T - 1, 5])
    mask = torch.zeros(B, T, dtype=torch.bool); mask[torch.arange(B), data["query_idx"]] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    heat = dec(bf, img, mask.cuda(), enc.predict(img), None)
    _, _, idx = hip_ops.decode_topk(heat)
    o = _oracle(data, dd, bd)
    err = (dec.last_logits.cpu() - o["logits"]).abs().max().item()
    same = (idx.cpu().long().sort(-1)[0] == o["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
    print(f"[T17 d2 {prec}] logits err {err:.3e} top-20 sets equal {same:.2f}")
    assert err <= 1e-3
    assert_identical_topk_sets(idx.cpu().long(), o, err, f"T17 d2 {prec}")
    assert dec.recast_count == 0          # the operand copy of the features arrived through features.attach


def test_views17_full_depth_default_mode(hip):
    """BASELINE configs[3]'s per-GPU shape at FULL depth (12 + 12 blocks, 1 query + 16 references: one 4352-token BETR sequence) in
    the default mode against the fp32 oracle (one pose: ~15 s of CPU): the 1e-3 bar and identical top-20 sets must hold at the longest
    sequence the configs name, not only at T = 6."""
    from boxdreamer_amd import _lib
    prec, B, T = _lib.DEFAULT_PREC, 1, 17
    enc, dec = _build(prec, 12, 12)
    data = synth.make_batch(seed=52, B=B, T=T)
    data["query_idx"] = torch.tensor([9])
    mask = torch.zeros(B, T, dtype=torch.bool); mask[0, 9] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    heat = dec(bf, img, mask.cuda(), enc.predict(img), None)
    _, _, idx = hip_ops.decode_topk(heat)
    o = _oracle(data, 12, 12)
    err = (dec.last_logits.cpu() - o["logits"]).abs().max().item()
    same = (idx.cpu().long().sort(-1)[0] == o["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
    print(f"[T17 full depth {prec}] logits err {err:.3e} top-20 sets equal {same:.2f}")
    REPORT[f"full_T17/{prec}"] = dict(logits=err, top20_sets_equal=same)
    assert err <= 1e-3
    assert_identical_topk_sets(idx.cpu().long(), o, err, f"T17 full depth {prec}")


@pytest.mark.parametrize("in_dtype", [torch.bfloat16, torch.float16])
def test_16bit_inputs_match_fp32_inputs(hip, in_dtype):
    """The dataset hands the model bf16 tensors; values are identical after the fp32 upcast, so results must be
    bit-identical to feeding fp32 (synth.make_batch pre-rounds through bf16, exactly representable in f16? no ->
    only bf16 is exact)."""
    a = _run("bf16", 1, 2, 2, 2, 7, torch.float32)
    b = _run("bf16", 1, 2, 2, 2, 7, in_dtype)
    if in_dtype == torch.bfloat16:
        assert torch.equal(a[2], b[2])
    else:
        assert (a[2] - b[2]).abs().max().item() < 0.05


@pytest.mark.parametrize("case", ["full_T2", "full_T6"])
def test_latency_forms_meet_the_bar_and_are_deterministic(hip, golden_dir, case):
    """`hip_latency: true` (ABI 9, bd_*_weights.latency_mode): the residual Linears of a one-pose call run split-K.  The opt-in trades the
    bit-identity with the throughput forms for latency -- the logits must still meet the 1e-3 bar against the CPU oracle and the REAL
    reference's fixture with identical top-20 sets, be deterministic, and the flag must actually change the launch forms (else this test
    measures nothing)."""
    prec = "f16c8_qk16"
    g = np.load(os.path.join(golden_dir, f"case_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    B, T, dd, bd, seed = meta["B"], meta["T"], meta["dino_depth"], meta["betr_depth"], meta["input_seed"]
    assert B == 1
    data, feats_t, logits_t, *_ = _run(prec, B, T, dd, bd, seed)            # throughput forms
    o = _oracle(data, dd, bd)
    enc, dec = _build(prec, dd, bd)
    enc.model.latency, dec.hip_latency = True, True
    mask = torch.zeros(B, T, dtype=torch.bool); mask[torch.arange(B), data["query_idx"]] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    outs = []
    for _ in range(2):
        heat = dec(bf, img, mask.cuda(), enc.predict(img), None)
        kp, kn, idx = hip_ops.decode_topk(heat)
        outs.append((dec.last_logits.cpu().clone(), kp.cpu().clone(), idx.cpu().long().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "the latency forms must be deterministic"
    logits, kp, idx = outs[0]
    e_logit = (logits - o["logits"]).abs().max().item()
    e_gold = np.abs(logits.reshape(B, -1)[:, ::7].numpy() - g["logits_strided"]).max()
    d_forms = (logits - logits_t).abs().max().item()
    print(f"[latency forms {case}] logits vs oracle {e_logit:.3e}, vs the reference fixture {e_gold:.3e}, vs the throughput forms {d_forms:.3e}")
    assert e_logit <= 1e-3 and e_gold <= 1e-3 + 1e-4
    assert 0.0 < d_forms <= 5e-4, "split-K must be in effect (different fp32 association) and stay inside the mode's margin"
    assert_identical_topk_sets(idx, o, e_logit, f"latency forms {case}")
    assert (kp - o["corners_px"]).abs().max().item() <= 224 / 20 * 2
    # switching back restores the throughput forms bit for bit
    enc.model.latency, dec.hip_latency = False, False
    dec(bf, img, mask.cuda(), enc.predict(img), None)
    assert torch.equal(dec.last_logits.cpu(), logits_t)


def test_batch_independence_and_determinism(hip):
    """Samples are independent units: a sample's result does not depend on its batch mates,
    and the path is run-to-run deterministic."""
    d2, _, l2, *_ = _run("bf16", 2, 3, 2, 2, 8)
    d2b, _, l2b, *_ = _run("bf16", 2, 3, 2, 2, 8)
    assert torch.equal(l2, l2b)
    enc, dec = _build("bf16", 2, 2)
    for b in range(2):
        img, bf = d2["images"][b:b + 1].cuda(), d2["bbox_feat"][b:b + 1].cuda()
        mask = torch.zeros(1, 3, dtype=torch.bool); mask[0, 2] = True
        dec(bf, img, mask.cuda(), enc.predict(img), None)
        assert torch.equal(dec.last_logits.cpu()[0], l2[b])


@pytest.mark.parametrize("prec", ["f16c8_qk16", "bf16x3"])
def test_query_view_position(hip, prec):
    """query_idx anywhere in the view list (reference samples it; betr.py:286-290,303)."""
    enc, dec = _build(prec, 2, 2)
    data = synth.make_batch(seed=8, B=2, T=3)
    data["query_idx"] = torch.tensor([0, 1])
    mask = torch.zeros(2, 3, dtype=torch.bool); mask[0, 0] = True; mask[1, 1] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    dec(bf, img, mask.cuda(), enc.predict(img), None)
    o = _oracle(data, 2, 2)
    assert (dec.last_logits.cpu() - o["logits"]).abs().max().item() <= 1e-3


@pytest.mark.parametrize("prec", ["f16c8_qk16", "bf16x3"])
@pytest.mark.parametrize("size,T,B", [(112, 3, 2), (84, 2, 1), (224, 2, 3), (98, 5, 3), (56, 1, 2), (224, 1, 1)])
def test_other_crop_sizes_and_view_counts(hip, size, T, B, prec):
    """img_size is a config value in the reference (configs/model/transformer.yaml:46); any multiple of 14 works:
    grid = size/14, DINO sequence = grid^2 + 5 (ragged tail tiles), BETR sequence = T * grid^2, decode over size^2."""
    dd, bd = 2, 2
    enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "synthetic_seed": 4321, "depth": dd,
                               "hip_precision": prec})
    dec = BETR(d_model=768, nhead=8, num_decoder_layers=bd, decoder_only=True, patch_size=14, img_size=size,
               diff_emb=False, nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True,
               patchify_rays=True, pose_representation="bb8", bbox_representation="heatmap", hip_precision=prec)
    dec.load_state_dict(synth.betr_state_dict(seed=1234, depth=bd), strict=True)
    dec = dec.cuda().eval()
    data = synth.make_batch(seed=31, B=B, T=T, size=size)
    data["query_idx"] = torch.arange(B) % T
    mask = torch.zeros(B, T, dtype=torch.bool); mask[torch.arange(B), data["query_idx"]] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    heat = dec(bf, img, mask.cuda(), enc.predict(img), None)
    kp, kn, idx = hip_ops.decode_topk(heat)
    o = orc.boxdreamer_forward(data, synth.betr_state_dict(1234, bd), synth.dino_state_dict(4321, dd))
    assert heat.shape == (B, 8, size, size)
    assert (dec.last_logits.cpu() - o["logits"]).abs().max().item() <= 1e-3
    assert (heat.cpu() - o["heat"]).abs().max().item() <= 1e-3
    assert torch.isfinite(data["bbox_feat"]).all()          # (round 5's synth rendered NaN maps at 56 px: an empty corner range)
    err = (dec.last_logits.cpu() - o["logits"]).abs().max().item()
    assert_identical_topk_sets(idx.cpu().long(), o, err, f"size{size}_T{T}_B{B}/{prec}")
    assert torch.equal(kp.cpu(), o["corners_px"]) or (kp.cpu() - o["corners_px"]).abs().max().item() <= 1e-3


def test_fp8_mode_restated_tolerance(hip):
    """BASELINE configs[4]: e4m3 Linears (block-scaled MFMA, unit scales, per-channel weight scales) + bf16 attention.
    e4m3 carries 3 mantissa bits, so the tolerance is RESTATED from measurement: the heatmap
    logits (std ~1) must stay within 1.0 max-abs / 0.2 rms of the fp32 oracle at full depth and the decoded corners
    within a few top-20 swaps."""
    data, feats, logits, heat, kp, kn, idx = _run("fp8", 1, 6, 12, 12, 11)
    o = _oracle(data, 12, 12)
    d = (logits - o["logits"])
    rep = {"logits_max_abs": d.abs().max().item(), "logits_rms": d.pow(2).mean().sqrt().item(),
           "logits_ref_rms": o["logits"].pow(2).mean().sqrt().item(),
           "feat_rms": (feats - o["rgb_feat"]).pow(2).mean().sqrt().item(),
           "corner_px_max": (kp - o["corners_px"]).abs().max().item()}
    print("[fp8 full_T6] " + json.dumps(rep))
    REPORT["full_T6/fp8"] = rep
    with open("gpurun_out/parity_report.json", "w") as f:
        json.dump(REPORT, f, indent=1)
    assert torch.isfinite(logits).all()
    assert rep["logits_max_abs"] <= 1.0 and rep["logits_rms"] <= 0.2


def test_fp8_mixed_policy(hip):
    """configs[4], usable form (round 4): "fp8_mixed" = the e4m3 class for the MLPs and DINOv2's QKV (2/3 of the Linear FLOPs), bf16
    for every proj, BETR's QKV, the adapter and the head (_lib.fp8_mixed_policy, through the per-Linear promotion bits).  Must be
    clearly closer to the fp32 oracle than all-e4m3, and stay inside ITS restated tolerance; then the decode question on PEAKED
    heatmaps (what a trained network emits; random weights give noise maps whose top-20 sets are unstable even in bf16): the same
    Gaussian corner peak added to the mode's logits and to the oracle's must decode to the same corner within 1 px."""
    res = {}
    for prec in ("fp8", "fp8_mixed", "bf16"):
        data, feats, logits, heat, kp, kn, idx = _run(prec, 1, 6, 12, 12, 11)
        o = _oracle(data, 12, 12)
        d = logits - o["logits"]
        same = (idx.sort(-1)[0] == o["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
        # peaked-heatmap decode: one Gaussian bump per corner map (sigma 3 px, 8 logits high) on top of both logit fields
        yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
        cx = torch.tensor([40.0, 180, 60, 150, 100, 120, 30, 200]); cy = torch.tensor([50.0, 60, 170, 160, 110, 40, 200, 120])
        bump = 8.0 * torch.exp(-((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / (2 * 3.0 ** 2))
        kp_m, _, _ = hip_ops.decode_topk((2 * torch.sigmoid(logits + bump[None]) - 1).cuda())
        kp_o, _, _ = hip_ops.decode_topk((2 * torch.sigmoid(o["logits"] + bump[None]) - 1).cuda())
        res[prec] = dict(logits_max_abs=d.abs().max().item(), logits_rms=d.pow(2).mean().sqrt().item(), top20_sets_equal=same,
                         corner_px_max=(kp - o["corners_px"]).abs().max().item(),
                         peaked_corner_px_max=(kp_m.cpu() - kp_o.cpu()).abs().max().item())
    print("[fp8 mixed] " + json.dumps(res))
    REPORT["fp8_mixed_T6"] = res
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fp8_mixed_report.json", "w") as f:
        json.dump(res, f, indent=1)
    # e4m3's error is spread evenly over the Linear types (CPU emulation, profiles/r4_fp8.md: any ONE type in e4m3 costs 0.05-0.065
    # rms), so keeping a third of the Linear FLOPs in bf16 buys ~20 %, not an order of magnitude
    assert res["fp8_mixed"]["logits_rms"] <= 0.9 * res["fp8"]["logits_rms"]
    assert res["fp8_mixed"]["logits_max_abs"] <= 0.6 and res["fp8_mixed"]["logits_rms"] <= 0.12
    assert res["fp8_mixed"]["peaked_corner_px_max"] <= 1.0 and res["fp8"]["peaked_corner_px_max"] <= 1.5


# ---------------------------------------------------------------- full-size properties (BASELINE configs[1]: B=32, T=6)

def _full(prec):
    enc, dec = _build(prec, 12, 12)
    small = synth.make_batch(seed=41, B=4, T=6)
    img = small["images"].repeat(8, 1, 1, 1, 1).to(torch.bfloat16)
    bf = small["bbox_feat"].repeat(8, 1, 1, 1, 1).to(torch.bfloat16)
    # make the 32 samples distinct: a different constant brightness / heat offset per repetition (exact in bf16)
    for r in range(8):
        img[4 * r:4 * r + 4] += 0.0625 * r
        bf[4 * r:4 * r + 4] *= (1.0 - 0.0625 * r)
    return enc, dec, img.cuda(), bf.cuda()


def _run_full(enc, dec, img, bf, qpos):
    B, T = img.shape[:2]
    mask = torch.zeros(B, T, dtype=torch.bool, device="cuda")
    mask[torch.arange(B), qpos] = True
    heat = dec(bf, img, mask, enc.predict(img), None)
    kp, _, idx = hip_ops.decode_topk(heat)
    return dec.last_logits.clone(), heat.clone(), kp.clone(), idx.clone().long()


@pytest.mark.parametrize("prec", ["f16c8_qk16", "bf16x3", "bf16"])
def test_full_size_properties(hip, prec):
    """The oracle cannot run B=32 x full depth in seconds, so the full-size step is checked through properties that do not
    depend on size:
      * determinism: the step is bit-reproducible;
      * sample independence: sample b of the batch-32 step == the same sample run alone (bit-exact: every kernel's
        per-row arithmetic order is independent of the batch);
      * reference-order invariance: BETR has no per-view embedding (betr.py:357-364 adds one 2-D table to every view) and
        attends jointly over all views, so permuting the REFERENCE views changes only the softmax summation order;
      * decode consistency: every decoded corner is the mean of its 20 selected pixels, and those pixels carry the 20
        largest heat values of the map (checked on the full-size output)."""
    enc, dec, img, bf = _full(prec)
    B, T = img.shape[:2]
    q = torch.full((B,), T - 1)
    l0, h0, kp0, idx0 = _run_full(enc, dec, img, bf, q)
    l1, *_ = _run_full(enc, dec, img, bf, q)
    assert torch.equal(l0, l1)
    for b in (0, 13, 31):
        lb, *_ = _run_full(enc, dec, img[b:b + 1], bf[b:b + 1], q[b:b + 1])
        assert torch.equal(lb[0], l0[b])
    perm = torch.tensor([3, 0, 4, 1, 2, 5])                          # references shuffled, query stays last
    lp, *_ = _run_full(enc, dec, img[:, perm].contiguous(), bf[:, perm].contiguous(), q)
    tol = 1e-3 if prec in STRICT else 1e-1
    assert (lp - l0).abs().max().item() <= tol
    # decode: top-20 really are the 20 largest, corners are their mean (x = idx % W, y = idx // W)
    Hh = h0.reshape(B * 8, -1)
    sel = Hh.gather(1, idx0.reshape(B * 8, 20))
    kth = Hh.topk(20, dim=1)[0][:, -1]
    assert (sel.min(1)[0] >= kth).all()
    xs, ys = (idx0 % 224).float().mean(-1), (idx0 // 224).float().mean(-1)
    assert (torch.stack([xs, ys], -1) - kp0).abs().max().item() <= 1e-3


def _full_config(prec, B, T, seed):
    """B distinct full-size samples of a BASELINE config: 4 seeded samples x (B / 4) exact-in-bf16 brightness / heat variations."""
    enc, dec = _build(prec, 12, 12)
    small = synth.make_batch(seed=seed, B=4, T=T)
    reps = B // 4
    img = small["images"].repeat(reps, 1, 1, 1, 1).to(torch.bfloat16)
    bf = small["bbox_feat"].repeat(reps, 1, 1, 1, 1).to(torch.bfloat16)
    for r in range(reps):
        img[4 * r:4 * r + 4] += 0.03125 * r
        bf[4 * r:4 * r + 4] *= (1.0 - 0.03125 * r)
    return enc, dec, img.cuda(), bf.cuda()


def test_config3_per_gpu_shard_properties(hip):
    """BASELINE configs[3] = 1 query + 16 refs, batch 256 over 8 GPUs: its PER-GPU shard (B = 32, T = 17, S = 4352 joint tokens per
    sample) at full depth in the default mode -- what every rank of the RCCL sweep runs; only the corner all-gather (tests/
    test_dist_gloo.py, bench.py --gpus) is missing from this 1-GPU form.  Size-independent properties: determinism, sample
    independence against B = 1 (bit-exact), decode consistency, finite in-range heatmaps."""
    enc, dec, img, bf = _full_config("f16c8_qk16", 32, 17, 43)
    B, T = img.shape[:2]
    assert (B, T) == (32, 17)
    q = torch.full((B,), T - 1)
    l0, h0, kp0, idx0 = _run_full(enc, dec, img, bf, q)
    l1, *_ = _run_full(enc, dec, img, bf, q)
    assert torch.equal(l0, l1)
    assert torch.isfinite(l0).all() and h0.abs().max().item() <= 1.0
    for b in (0, 17, 31):
        lb, *_ = _run_full(enc, dec, img[b:b + 1], bf[b:b + 1], q[b:b + 1])
        assert torch.equal(lb[0], l0[b]), b
    assert not torch.equal(l0[0], l0[4])                              # the samples really differ
    Hh = h0.reshape(B * 8, -1)
    sel = Hh.gather(1, idx0.reshape(B * 8, 20))
    assert (sel.min(1)[0] >= Hh.topk(20, dim=1)[0][:, -1]).all()
    xs, ys = (idx0 % 224).float().mean(-1), (idx0 // 224).float().mean(-1)
    assert (torch.stack([xs, ys], -1) - kp0).abs().max().item() <= 1e-3


def test_config4_fp8_batch64_properties(hip):
    """BASELINE configs[4] = fp8 (e4m3) Linears, 5 refs, batch 64 on one GPU, at full size: determinism, sample independence against
    B = 1 (bit-exact: the e4m3 kernels' per-row arithmetic does not depend on the batch either), decode consistency, and the mode's
    RESTATED tolerance (test_fp8_mode_restated_tolerance) against the split-f16 mode on the same device at full size -- the oracle
    cannot run 64 full-depth poses in seconds."""
    enc, dec, img, bf = _full_config("fp8", 64, 6, 47)
    B, T = img.shape[:2]
    assert (B, T) == (64, 6)
    q = torch.full((B,), T - 1)
    l0, h0, kp0, idx0 = _run_full(enc, dec, img, bf, q)
    l1, *_ = _run_full(enc, dec, img, bf, q)
    assert torch.equal(l0, l1) and torch.isfinite(l0).all()
    for b in (0, 33, 63):
        lb, *_ = _run_full(enc, dec, img[b:b + 1], bf[b:b + 1], q[b:b + 1])
        assert torch.equal(lb[0], l0[b]), b
    Hh = h0.reshape(B * 8, -1)
    sel = Hh.gather(1, idx0.reshape(B * 8, 20))
    assert (sel.min(1)[0] >= Hh.topk(20, dim=1)[0][:, -1]).all()
    del enc, dec
    enc3, dec3 = _build("f16x3", 12, 12)
    lr, *_ = _run_full(enc3, dec3, img[:8], bf[:8], q[:8])
    d = l0[:8] - lr
    assert d.abs().max().item() <= 1.0 and d.pow(2).mean().sqrt().item() <= 0.2


# ---------------------------------------------------------------- the default mode and its margin to the bar

def test_default_precision_is_the_parity_meeting_mode(hip, golden_dir):
    """a maintainer who applies INTEGRATION.md's 3-line patch and names no precision must get the mode that meets
    north_star's bar (logits <= 1e-3 vs the fp32 CPU forward, identical top-20 sets) -- bf16 is the explicit opt-in."""
    from boxdreamer_amd import _lib
    assert _lib.DEFAULT_PREC == "f16c8_qk16"
    enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "synthetic_seed": 4321, "depth": 12})
    dec = BETR(d_model=768, nhead=8, num_decoder_layers=12, decoder_only=True, patch_size=14, img_size=224,
               diff_emb=False, nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True,
               patchify_rays=True, pose_representation="bb8", bbox_representation="heatmap")
    assert enc.prec == dec.hip_precision == "f16c8_qk16"
    dec.load_state_dict(synth.betr_state_dict(seed=1234, depth=12), strict=True)
    dec = dec.cuda().eval()
    g = np.load(os.path.join(golden_dir, "case_full_T6.npz"))
    meta = json.loads(str(g["meta"]))
    data = synth.make_batch(seed=meta["input_seed"], B=meta["B"], T=meta["T"])
    mask = torch.zeros(meta["B"], meta["T"], dtype=torch.bool); mask[torch.arange(meta["B"]), data["query_idx"]] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    with torch.inference_mode():                       # Lightning's test / predict loops run the model here
        heat = dec(bf, img, mask.cuda(), enc.predict(img), None)
    assert dec.recast_count == 0
    e_gold = np.abs(dec.last_logits.cpu().reshape(meta["B"], -1)[:, ::7].numpy() - g["logits_strided"]).max()
    assert e_gold <= 1e-3, e_gold
    _, _, idx = hip_ops.decode_topk(heat)
    assert np.array_equal(np.sort(idx.cpu().numpy().reshape(-1, 20), -1), g["topk_idx_sorted"].reshape(-1, 20))


MARGIN_INPUT_SEEDS = (101, 102, 103, 104, 105, 106, 107, 108)
MARGIN_WEIGHT_SEEDS = ((1234, 4321), (777, 888))
_MARGIN_ORACLE = {}      # (weight seeds, input seed) -> the fp32 CPU oracle's (logits, top-20 indices): computed once, used by both modes


def _margin_oracle(ws_b, ws_d, datas, bsd, dsd):
    """Per-pose oracle outputs for one weight set: ONE batched fp32 CPU forward over the eight poses (samples are independent units)."""
    key = (ws_b, ws_d)
    if key not in _MARGIN_ORACLE:
        batch = {k: torch.cat([d[k] for d in datas]) for k in ("images", "bbox_feat", "query_idx")}
        o = orc.boxdreamer_forward(batch, bsd, dsd)
        _MARGIN_ORACLE[key] = [{"logits": o["logits"][i:i + 1].clone(), "topk_idx": o["topk_idx"][i:i + 1].clone()} for i in range(len(datas))]
    return _MARGIN_ORACLE[key]


@pytest.mark.parametrize("prec", ["f16c8_qk16"])
def test_strict_mode_margin_over_seeds(hip, prec):
    """the strict mode's distance to the 1e-3 bar on MORE than a handful of seeds -- 8 input seeds x 2 weight
    seeds at FULL depth, T = 6, each pose against the fp32 CPU oracle (~1 s each).  Every pose must meet the bar with identical
    top-20 sets; the distribution goes to gpurun_out/strict_margin_<mode>.json (and from there to profiles/).  The bound asserted is
    5e-4 -- half the bar ("<= 5e-4 on the 16-pose margin test")."""
    errs, sets_equal, near_ties = [], [], []
    for ws_b, ws_d in MARGIN_WEIGHT_SEEDS:
        enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "synthetic_seed": ws_d, "depth": 12, "hip_precision": prec})
        dec = BETR(d_model=768, nhead=8, num_decoder_layers=12, decoder_only=True, patch_size=14, img_size=224,
                   diff_emb=False, nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True,
                   patchify_rays=True, pose_representation="bb8", bbox_representation="heatmap", hip_precision=prec)
        bsd, dsd = synth.betr_state_dict(seed=ws_b, depth=12), synth.dino_state_dict(seed=ws_d, depth=12)
        dec.load_state_dict(bsd, strict=True)
        dec = dec.cuda().eval()
        B, T = len(MARGIN_INPUT_SEEDS), 6
        datas = [synth.make_batch(seed=sd, B=1, T=T) for sd in MARGIN_INPUT_SEEDS]
        for i, d in enumerate(datas):
            d["query_idx"] = torch.tensor([(i * 5) % T])             # query position varies over the seeds
        img = torch.cat([d["images"] for d in datas]).cuda()
        bf = torch.cat([d["bbox_feat"] for d in datas]).cuda()
        q = torch.cat([d["query_idx"] for d in datas])
        mask = torch.zeros(B, T, dtype=torch.bool); mask[torch.arange(B), q] = True
        heat = dec(bf, img, mask.cuda(), enc.predict(img), None)
        _, _, idx = hip_ops.decode_topk(heat)
        logits, idx = dec.last_logits.cpu(), idx.cpu().long()
        oracle = _margin_oracle(ws_b, ws_d, datas, bsd, dsd)
        for i, d in enumerate(datas):
            o = oracle[i]
            e = (logits[i] - o["logits"][0]).abs().max().item()
            errs.append(e)
            eq = (idx[i].sort(-1)[0] == o["topk_idx"][0].sort(-1)[0]).all(-1)          # per corner map
            sets_equal.append(bool(eq.all()))
            # a differing top-20 set is legitimate only where the ORACLE's own 20th / 21st logits are closer than the error bound
            # (random-weight heatmaps are noise: near-ties exist); anything else is a real decode difference
            for c in (~eq).nonzero().flatten().tolist():
                top = o["logits"][0, c].flatten().topk(21)[0]
                gap = (top[19] - top[20]).item()
                near_ties.append(gap)
                assert gap <= 2.0 * e, (i, c, gap, e)
    rep = {"mode": prec, "poses": len(errs), "logits_max_abs_err": {"max": max(errs), "median": float(np.median(errs)), "min": min(errs),
           "all": [round(e, 7) for e in errs]}, "top20_sets_equal": sum(sets_equal),
           "oracle_gap_20th_21st_where_sets_differ": near_ties, "bar": 1e-3, "margin_x": 1e-3 / max(errs),
           "weight_seeds": MARGIN_WEIGHT_SEEDS, "input_seeds": MARGIN_INPUT_SEEDS}
    print("[strict margin] " + json.dumps(rep))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/strict_margin_{prec}.json", "w") as f:
        json.dump(rep, f, indent=1)
    assert max(errs) <= 5e-4, errs
    # every differing set was checked above to sit at an oracle near-tie (gap <= 2 x that pose's error); anything else failed there


# ---------------------------------------------------------------- operand-range robustness

def _run_with(prec, bsd, dsd, data, depth):
    enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "state_dict": dsd, "hip_precision": prec})
    enc.to_device("cuda")
    dec = BETR(d_model=768, nhead=8, num_decoder_layers=depth, decoder_only=True, patch_size=14, img_size=224,
               diff_emb=False, nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True,
               patchify_rays=True, pose_representation="bb8", bbox_representation="heatmap", hip_precision=prec)
    dec.load_state_dict(bsd, strict=True)
    dec = dec.cuda().eval()
    B, T = data["images"].shape[:2]
    mask = torch.zeros(B, T, dtype=torch.bool); mask[torch.arange(B), data["query_idx"]] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    feats = enc.predict(img)
    heat = dec(bf, img, mask.cuda(), feats, None)
    _, _, idx = hip_ops.decode_topk(heat)
    return feats.cpu(), dec.last_logits.cpu(), idx.cpu().long()


@pytest.mark.parametrize("prec", ["f16c8_qk16", "f16c8", "bf16x3"])
@pytest.mark.parametrize("case", ["full_T2", "full_T6"])
def test_range_stress_function_preserving_rescale(hip, golden_dir, prec, case):
    """Every range limit of the strict operand classes, with the network's FUNCTION unchanged (synth.rescale_function_preserving:
    power-of-two gains moved between a producer and its only consumer, so the fp32 forward is bit-identical -- asserted in
    tests/test_oracle_golden.py -- and the REFERENCE's own fixture for the plain weights is the expected output): LayerNorm outputs
    reach ~1000 (beyond e4m3's 448: round 2 clamped the whole activation there, which oracle/numerics_sim.py shows is catastrophic:
    0.27 on the logits), the consuming weight columns sit at ~1e-4 next to ordinary ones in the same tensor (per-tensor w_qexp, f16
    subnormals), attention values are x64, DINOv2 q / k features x32 / 32.  The 1e-3 bar must hold unchanged."""
    # T = 2: the fixture the REAL reference wrote on the rescaled weights themselves (oracle/make_golden.py:robustness_cases);
    # T = 6: the plain-weights fixture (same function)
    g = np.load(os.path.join(golden_dir, "case_rescaled_T2.npz" if case == "full_T2" else f"case_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    B, T, seed = meta["B"], meta["T"], meta["input_seed"]
    dsd, bsd = synth.rescale_function_preserving(synth.dino_state_dict(4321, 12), synth.betr_state_dict(1234, 12))
    data = synth.make_batch(seed=seed, B=B, T=T)
    feats, logits, idx = _run_with(prec, bsd, dsd, data, 12)
    assert feats.abs().max().item() > 448.0                       # the stress is real: operands beyond e4m3's range
    e_gold = float(np.abs(logits.reshape(B, -1)[:, ::7].numpy() - g["logits_strided"]).max())
    print(f"[range stress {case} {prec}] logits vs the reference's fixture: {e_gold:.3e}  (max |feature| {feats.abs().max().item():.0f})")
    REPORT[f"range_stress_{case}/{prec}"] = dict(logits_vs_golden=e_gold)
    assert e_gold <= 1e-3, e_gold
    assert np.array_equal(np.sort(idx.numpy().reshape(-1, 20), -1), g["topk_idx_sorted"].reshape(-1, 20))


@pytest.mark.parametrize("gain", [0.5, 1.0])
def test_trained_like_outliers(hip, gain):
    """Trained-like statistics that DO change the function (synth.*_outliers: massive-activation channels, LayerNorm gain
    outliers, MLP hidden units with pre-activations in the hundreds, q / k gain outliers).  At gain 1.0 single A-operand elements
    reach ~800 and the heatmap logits grow to rms ~5 / max ~25 -- the network amplifies every rounding: the CPU's own fp32
    re-association noise is 5e-4 there (oracle/numerics_sim.py --outliers 1), so the 1e-3 ABSOLUTE bar is RESTATED for this case
    relative to the logit scale: err <= 1e-3 x max(1, max|logits|); it must be finite (no saturation blow-up: round 2's clamp gave
    9.6 here), within 2x (+1e-3) of the most precise GPU mode (split-bf16 everywhere: what this conditioning allows), and decode
    the same corners."""
    depth, T = 12, 2
    bsd, dsd = synth.betr_state_dict_outliers(1234, depth, gain), synth.dino_state_dict_outliers(4321, depth, gain)
    data = synth.make_batch(seed=11, B=1, T=T)
    o = orc.boxdreamer_forward(data, bsd, dsd)
    rms = o["logits"].pow(2).mean().sqrt().item()
    res = {}
    for prec in ("f16c8_qk16", "bf16x3_attn_x3"):
        _, logits, idx = _run_with(prec, bsd, dsd, data, depth)
        assert torch.isfinite(logits).all()
        same = (idx.sort(-1)[0] == o["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
        res[prec] = ((logits - o["logits"]).abs().max().item(), same)
    print(f"[outliers gain {gain}] logits rms {rms:.2f} max {o['logits'].abs().max().item():.1f}: " + json.dumps(res))
    REPORT[f"outliers_g{gain}"] = dict(logit_rms=rms, **{k: v[0] for k, v in res.items()})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.json", "w") as f:
        json.dump(REPORT, f, indent=1)
    e_strict, e_x3 = res["f16c8_qk16"][0], res["bf16x3_attn_x3"][0]
    if gain == 0.5:            # the REAL reference's own output on these weights (tests/golden/case_outliers_g0.5_T2.npz)
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "case_outliers_g0.5_T2.npz"))
        assert np.abs(o["logits"].reshape(1, -1)[:, ::7].numpy() - g["logits_strided"]).max() <= 5e-4
        _, lg, _ = _run_with("f16c8_qk16", bsd, dsd, data, depth)
        e_gold = float(np.abs(lg.reshape(1, -1)[:, ::7].numpy() - g["logits_strided"]).max())
        print(f"[outliers gain 0.5] strict mode vs the reference's fixture: {e_gold:.3e}")
        assert e_gold <= 1e-3 * max(1.0, float(g["logits_absmax"]))
    assert e_strict <= 1e-3 * max(1.0, o["logits"].abs().max().item()), (e_strict, rms)
    assert e_strict <= 2.0 * e_x3 + 1e-3, (e_strict, e_x3)
    assert res["f16c8_qk16"][1] >= 0.85
