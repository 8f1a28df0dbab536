"""GPU: sub-batch lanes (include/boxdreamer_hip.h, ABI v6: bd_encoder_forward_lanes / bd_decoder_forward_lanes).

One batch run as 2-4 contiguous sub-batches on as many streams must give BIT-identical outputs to the plain one-stream form:
samples are independent on this path (the reference loops per sample, prediction_utils.py:63-101; BETR attends within a sample,
betr.py:282-296) and a row's result does not depend on the launch geometry.  Covered: every operand class's feature hand-off
(one plane, two 16-bit planes, F16C8's byte plane, e4m3 bytes), uneven splits, a promoted adapter (features handed over in the
promoted class), non-trailing query views, HIP-graph capture of the laned form, use from a non-default stream.
"""
import pytest
import torch

from boxdreamer_amd import _lib, hip_ops, synth
from boxdreamer_amd.betr import BETR
from boxdreamer_amd.encoder import DinoV2Wrapper

pytestmark = pytest.mark.gpu


def _build(prec, depth=2):
    enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "synthetic_seed": 4321, "depth": depth, "hip_precision": prec})
    dec = BETR(d_model=768, nhead=8, num_decoder_layers=depth, decoder_only=True, patch_size=14, img_size=224,
               diff_emb=False, nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True,
               patchify_rays=True, pose_representation="bb8", bbox_representation="heatmap", hip_precision=prec)
    dec.load_state_dict(synth.betr_state_dict(seed=1234, depth=depth), strict=True)
    return enc, dec.cuda().eval()


def _same(a, b, what):
    for x, y, name in zip(a, b, ("feats32", "feats16", "logits", "heat", "corners", "top-20 indices")):
        if x is None and y is None:
            continue
        xb, yb = x.contiguous().view(torch.uint8), y.contiguous().view(torch.uint8)
        assert torch.equal(xb, yb), f"{what}: {name} differs between the laned and the plain form"


@pytest.mark.parametrize("prec", ["bf16", "f16c8_qk16", "bf16x3", "f16x3", "fp8"])
@pytest.mark.parametrize("B,lanes", [(4, 2), (5, 2), (5, 3), (7, 4)])
def test_lanes_bit_identical(hip, prec, B, lanes):
    if (B, lanes) not in ((4, 2), (5, 3)) and prec not in ("bf16", "f16c8_qk16"):
        pytest.skip("uneven / four-lane splits are covered in the two benched modes")
    enc, dec = _build(prec)
    T = 3
    data = synth.make_batch(seed=11 + B, B=B, T=T)
    qi = torch.arange(B) % T                                   # query views at different positions
    mask = torch.zeros(B, T, dtype=torch.bool); mask[torch.arange(B), qi] = True
    img, bf = data["images"].to(torch.bfloat16).cuda(), data["bbox_feat"].to(torch.bfloat16).cuda()
    from boxdreamer_amd import features

    def run(l):
        enc.model.lanes = l; dec.hip_lanes = l
        feats = enc.predict(img)
        opc = features.tag_of(feats)[0].clone()
        if _lib.operand_prec(prec) == _lib.PREC_F16C8:        # plane 1 holds ONE byte per element: only its first half is written
            flat = opc.view(torch.uint8).reshape(2, -1)
            opc = torch.cat([flat[0], flat[1, : flat.shape[1] // 2]])
        heat = dec(bf, img, mask.cuda(), feats, None)
        kp, _, idx = hip_ops.decode_topk(heat)
        torch.cuda.synchronize()
        return feats.clone(), opc, dec.last_logits.clone(), heat.clone(), kp.clone(), idx.clone()

    plain, laned = run(1), run(lanes)
    _same(plain, laned, f"{prec} B={B} lanes={lanes}")
    assert torch.isfinite(plain[2]).all()


def test_lanes_with_a_promoted_adapter(hip):
    """Features handed over in the PROMOTED class (split-f16 planes) while the base class is F16C8: the lane slices of the hand-off
    buffer follow the class the decoder reads, not the base class."""
    prec = "f16c8_qk16"
    enc, dec = _build(prec)
    dec.hip_promote_misc = _lib.PROMOTE_ADAPTER_FC1 | _lib.PROMOTE_ADAPTER_FC2
    dec.hip_promote = [_lib.PROMOTE_QKV | _lib.PROMOTE_FC1 | _lib.PROMOTE_FC2, 0]
    enc.model.feats_prec = dec.feats_class(prec)
    enc.model.promote = [_lib.PROMOTE_PROJ, _lib.PROMOTE_FC1 | _lib.PROMOTE_FC2]
    B, T = 4, 2
    data = synth.make_batch(seed=21, B=B, T=T)
    mask = torch.zeros(B, T, dtype=torch.bool); mask[:, T - 1] = True
    img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
    outs = []
    for l in (1, 2):
        enc.model.lanes = l; dec.hip_lanes = l
        feats = enc.predict(img)
        heat = dec(bf, img, mask.cuda(), feats, None)
        torch.cuda.synchronize()
        outs.append((feats.clone(), dec.last_logits.clone(), heat.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


def test_lanes_inside_a_captured_graph_and_on_a_side_stream(hip):
    from boxdreamer_amd.graph import GraphedPath
    prec = "bf16"
    enc, dec = _build(prec)
    B, T = 4, 3
    data = synth.make_batch(seed=31, B=B, T=T)
    img, bf = data["images"].to(torch.bfloat16).cuda(), data["bbox_feat"].to(torch.bfloat16).cuda()
    mask = torch.zeros(B, T, dtype=torch.bool, device="cuda"); mask[:, T - 1] = True
    enc.model.lanes = 1; dec.hip_lanes = 1
    feats = enc.predict(img)
    heat1 = dec(bf, img, mask, feats, None).clone()
    kp1 = hip_ops.decode_topk(heat1)[0].clone()
    torch.cuda.synchronize()
    # a caller on its own stream (the library forks from / joins into THAT stream)
    enc.model.lanes = 2; dec.hip_lanes = 2
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        heat_s = dec(bf, img, mask, enc.predict(img), None)
        kp_s = hip_ops.decode_topk(heat_s)[0]
    s.synchronize()
    assert torch.equal(heat_s, heat1) and torch.equal(kp_s, kp1)
    torch.cuda.current_stream().wait_stream(s)
    # captured: the side streams join the capture through the fork event
    g = GraphedPath(enc, dec, B, T, 224, torch.bfloat16, "cuda")
    for rep in range(3):
        heat_g, kp_g, _, _ = g(img, bf)
        torch.cuda.synchronize()
        assert torch.equal(heat_g, heat1) and torch.equal(kp_g, kp1), f"replay {rep}"
    # other inputs through the same graph
    data2 = synth.make_batch(seed=32, B=B, T=T)
    img2, bf2 = data2["images"].to(torch.bfloat16).cuda(), data2["bbox_feat"].to(torch.bfloat16).cuda()
    heat_g2 = g(img2, bf2)[0].clone()
    torch.cuda.synchronize()
    del g
    enc.model.lanes = 1; dec.hip_lanes = 1
    heat2 = dec(bf2, img2, mask, enc.predict(img2), None)
    torch.cuda.synchronize()
    assert torch.equal(heat_g2, heat2)


def test_lanes_argument_checks(hip):
    lib = _lib.load()
    enc, dec = _build("bf16")
    enc.model.lanes = 9
    with pytest.raises(ValueError):
        enc.predict(torch.zeros(2, 3, 224, 224, device="cuda"))
    enc.model.lanes = "auto"
    assert _lib.resolve_lanes("auto", 6, 1) == 1 and _lib.resolve_lanes("auto", 192, 32) == 2 and _lib.resolve_lanes(4, 12, 2) == 2
    # workspace of the laned form is what the forward checks against
    pk = enc.model._weights(224, "bf16")
    need1 = lib.bd_encoder_workspace_bytes(pk.struct, 6, _lib.prec_id("bf16"))
    need2 = lib.bd_encoder_workspace_bytes_lanes(pk.struct, 6, _lib.prec_id("bf16"), 2)
    assert need1 > 0 and need2 > 0 and 0 <= lib.bd_encoder_workspace_bytes_lanes(pk.struct, 6, _lib.prec_id("bf16"), 1) - need1 < 256
    img = torch.zeros(6, 3, 224, 224, device="cuda")
    ws = torch.empty(need2 - 512, dtype=torch.uint8, device="cuda")
    f32 = torch.empty(6 * 256, 768, device="cuda")
    rc = lib.bd_encoder_forward_lanes(pk.struct, _lib.ptr(img), _lib.dtype_id(img), 6, 224, _lib.ptr(f32), None, 0, _lib.ptr(ws),
                                      ws.numel(), _lib.prec_id("bf16"), 2, _lib.stream())
    with pytest.raises(_lib.HipLibraryError, match="BD_ERR_WORKSPACE"):
        _lib.check(rc, "bd_encoder_forward_lanes")


def test_facade_runs_a_large_batch_as_two_lanes_and_says_so(hip):
    """`BoxDreamer.forward` (BoxDreamerModel.py:112-191) on a batch of >= 24 (sample, view) images: two lanes by default, recorded in
    the output dict, every output equal to the `hip_lanes: 1` run."""
    import copy
    from boxdreamer_amd.model import BoxDreamer
    from test_gpu_facade import _config_with
    B, T = 11, 6
    data = synth.make_batch(seed=41, B=B, T=T, dtype=torch.bfloat16)
    data["query_idx"] = torch.arange(B) % T
    outs = []
    for lanes in ("auto", 1):
        cfg = _config_with("f16c8_qk16")
        if lanes != "auto":
            cfg["modules"]["decoder"]["hip_lanes"] = lanes
            cfg["modules"]["encoder"]["dino"]["cfg"]["hip_lanes"] = lanes
        model = BoxDreamer(cfg)
        model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
        model = model.cuda().eval()
        dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in copy.deepcopy(data).items()}
        with torch.inference_mode():
            out = model(dev)
        torch.cuda.synchronize()
        assert out["hip_precision"]["sub_batch_lanes"] == (2 if lanes == "auto" else 1)
        outs.append({k: out[k].clone() for k in ("pred_bbox", "pred_corners_px", "pred_poses", "regression_boxes")})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_one_thread_captures_while_another_enqueues_laned_calls(hip):
    """The lanes' side streams / events belong to the calling HOST THREAD (round 6; process-global before: a thread capturing next to a
    thread enqueueing eagerly pulled the other thread's side-stream work into its capture).  Thread B runs eager two-lane
    forwards in a loop while the main thread warms up, CAPTURES and replays the two-lane step of another module pair; both must give the
    bits of their single-threaded runs."""
    import threading
    from boxdreamer_amd.graph import GraphedPath
    prec, B, T = "f16c8_qk16", 4, 6
    encA, decA = _build(prec)
    encB, decB = _build(prec)
    for e, d in ((encA, decA), (encB, decB)):
        e.model.lanes, d.hip_lanes = 2, 2
    dA, dB = synth.make_batch(seed=31, B=B, T=T), synth.make_batch(seed=32, B=B, T=T)
    mask = torch.zeros(B, T, dtype=torch.bool); mask[:, T - 1] = True
    mask = mask.cuda()
    imgA, bfA = dA["images"].to(torch.bfloat16).cuda(), dA["bbox_feat"].to(torch.bfloat16).cuda()
    imgB, bfB = dB["images"].to(torch.bfloat16).cuda(), dB["bbox_feat"].to(torch.bfloat16).cuda()

    def eager(enc, dec, img, bf):
        heat = dec(bf, img, mask, enc.predict(img), None)
        return dec.last_logits.clone(), heat.clone()
    refA, refB = eager(encA, decA, imgA, bfA), eager(encB, decB, imgB, bfB)
    torch.cuda.synchronize()
    stop, errs, runs = threading.Event(), [], [0]

    def worker():
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                while not stop.is_set():
                    got = eager(encB, decB, imgB, bfB)
                    s.synchronize()
                    if not (torch.equal(got[0], refB[0]) and torch.equal(got[1], refB[1])):
                        errs.append("thread B's eager laned forward changed while thread A captured")
                        return
                    runs[0] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
    th = threading.Thread(target=worker)
    th.start()
    try:
        while runs[0] < 2 and not errs:          # B is running before A starts to capture
            torch.cuda.current_stream().synchronize()
        g = GraphedPath(encA, decA, B, T, 224, torch.bfloat16, "cuda", capture_error_mode="thread_local")
        for _ in range(5):
            heat, kp, kn, _ = g(imgA, bfA)
            torch.cuda.current_stream().synchronize()
            assert torch.equal(heat, refA[1]), "the captured two-lane step differs from the single-threaded run"
    finally:
        stop.set()
        th.join(60)
    assert not errs, errs
    assert runs[0] >= 3
