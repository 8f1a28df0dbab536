"""GPU: the drop-in boundary `BoxDreamer(config).forward(data) -> data` (BoxDreamerModel.py:112-191) end to end,
constructed from the reference's own config layout (configs/model/transformer.yaml:10-71)."""
import copy

import pytest
import torch

from boxdreamer_amd import synth
from boxdreamer_amd.model import BoxDreamer
from oracle import boxdreamer_oracle as orc

pytestmark = pytest.mark.gpu


STRICT_DEFAULT = "f16c8_qk16"       # the facade's default mode (boxdreamer_amd._lib.DEFAULT_PREC)


def _config(prec, depth=2):
    """prec None: no `hip_precision` key anywhere -- what the reference's own YAML gives a maintainer."""
    cfg = _config_with(prec if prec is not None else "x", depth)
    if prec is None:
        del cfg["modules"]["decoder"]["hip_precision"]
        del cfg["modules"]["encoder"]["dino"]["cfg"]["hip_precision"]
    return cfg


def _config_with(prec, depth=2):
    """The constructor's `config`: the reference's OWN YAML resolved to data (tests/golden/model_modules_config.json, written by
    oracle/make_model_config.py from configs/model/transformer.yaml:10-71 + configs/test.yaml:8-24 -- not a hand-typed dict), with the
    three things a test must add on top: the layer count of the reduced-depth cases, seeded synthetic DINOv2 weights in place of the hub
    download, and the HIP precision mode."""
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_modules_config.json")
    mods = copy.deepcopy(json.load(open(path))["modules"])
    mods["decoder"].update(num_decoder_layers=depth, hip_precision=prec)
    mods["encoder"]["dino"]["cfg"].update(synthetic_seed=4321, depth=depth, hip_precision=prec)
    return {"modules": mods}


@pytest.mark.parametrize("prec", [None, "bf16x3"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_dict_contract(hip, dtype, prec):
    model = BoxDreamer(_config(prec))
    if prec is None:
        assert model.decoder.hip_precision == model.rgb_encoder.prec == STRICT_DEFAULT
    # checkpoints carry the decoder under "decoder." (Lightning adds "BoxDreamer." on top, demo.py:564-573 strips it)
    sd = {"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    model = model.cuda().eval()
    B, T = 2, 3
    data = synth.make_batch(seed=8, B=B, T=T, dtype=dtype)
    data["query_idx"] = torch.tensor([2, 0])
    keys_in = set(data)
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    inputs = copy.deepcopy({k: v.clone() for k, v in dev.items()})
    if prec is None:
        with torch.inference_mode():              # Lightning's test / validate / predict loops (used to crash here)
            out = model(dev)
    else:
        out = model(dev)
    assert out is dev
    for k in ("camera_mask", "pred_bbox", "regression_boxes", "pred_poses", "pred_intrinsics"):
        assert k in out
    for k in keys_in:                                   # inputs are not modified
        assert torch.equal(out[k], inputs[k])
    cm = out["camera_mask"].cpu()
    assert cm.dtype == torch.bool and cm.sum(1).tolist() == [1, 1] and cm[0, 2] and cm[1, 0]
    assert out["pred_bbox"].dtype == dtype and out["pred_bbox"].shape == (B, T, 8, 224, 224)
    assert torch.equal(out["pred_bbox"][~out["camera_mask"]], inputs["bbox_feat"][~out["camera_mask"]])
    o = orc.boxdreamer_forward({**data, "images": data["images"].float(), "bbox_feat": data["bbox_feat"].float()},
                               synth.betr_state_dict(1234, 2), synth.dino_state_dict(4321, 2))
    assert (model.decoder.last_logits.cpu() - o["logits"]).abs().max().item() <= 1e-3
    tol = 1e-4 if dtype == torch.float32 else 8e-3
    assert (out["pred_bbox"][out["camera_mask"]].float().cpu() - o["heat"]).abs().max().item() <= tol
    assert (out["pred_corners_px"].cpu() - o["corners_px"]).abs().max().item() <= 224 / 20 * 2
    rb = out["regression_boxes"].float().cpu()
    assert (rb[cm] - o["corners_norm"]).abs().max().item() <= (0.2 if dtype == torch.float32 else 0.25)
    assert torch.equal(rb[~cm], data["bbox_proj_crop"].float()[~cm])
    pp = out["pred_poses"].float().cpu()
    assert pp.shape == (B, T, 4, 4) and torch.isfinite(pp).all()
    assert torch.equal(pp[~cm], data["poses"].float()[~cm])             # reference views keep their GT pose
    # train mode: no evaluation post-process (BoxDreamerModel.py:166-185)
    model.train()
    dev2 = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    out2 = model(dev2)
    assert "regression_boxes" not in out2 and torch.equal(out2["pred_poses"], dev2["poses"])


@pytest.mark.parametrize("graph", [False, True])
def test_facade_latency_opt_in(hip, graph):
    """config["modules"]["hip_latency"] = True (the reference demo's per-frame call, src/demo/demo.py:1501-1514): the modules' latency forms
    (ABI 9, split-K residual Linears) behind the same forward(dict) -> dict; within the bar against the CPU oracle, deterministic, eager ==
    HIP graph; the default (flag absent) stays the throughput forms."""
    cfg = _config_with(STRICT_DEFAULT)
    assert BoxDreamer(copy.deepcopy(cfg)).decoder.hip_latency is False
    cfg["modules"]["hip_latency"] = True
    cfg["modules"]["hip_graph"] = graph
    model = BoxDreamer(cfg)
    assert model.decoder.hip_latency is True and model.rgb_encoder.model.latency is True
    model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
    model = model.cuda().eval()
    data = synth.make_batch(seed=8, B=1, T=6)
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    out1 = model(dict(dev))
    l1, c1 = model.decoder.last_logits.clone(), out1["pred_corners_px"].clone()
    out2 = model(dict(dev))
    assert torch.equal(model.decoder.last_logits, l1) and torch.equal(out2["pred_corners_px"], c1)
    o = orc.boxdreamer_forward(data, synth.betr_state_dict(1234, 2), synth.dino_state_dict(4321, 2))
    assert (l1.cpu() - o["logits"]).abs().max().item() <= 1e-3
    assert (c1.cpu() - o["corners_px"]).abs().max().item() <= 224 / 20 * 2
    assert torch.isfinite(out2["pred_poses"]).all()
    # the throughput forms on the same input differ in fp32 association only
    ref = BoxDreamer(_config_with(STRICT_DEFAULT))
    ref.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
    ref = ref.cuda().eval()
    ref(dict(dev))
    d = (ref.decoder.last_logits - l1).abs().max().item()
    assert 0.0 < d <= 5e-4, d


def test_unsupported_configs_raise():
    cfg = _config("bf16")
    cfg["modules"]["use_tracking"] = True
    with pytest.raises(NotImplementedError):
        BoxDreamer(cfg)
    cfg = _config("bf16")
    cfg["modules"]["encoder"]["name"] = "resnet"
    with pytest.raises(NotImplementedError):
        BoxDreamer(cfg)


def test_reference_feature_cache_is_bit_identical(hip):
    """"next" row f1: encoding the references once and only the query per pose gives the same bits."""
    from boxdreamer_amd.cache import RefFeatureCache
    for prec in ("bf16", "bf16x3", "f16c8_qk16", "f16c8"):
        model = BoxDreamer(_config(prec))
        model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
        model = model.cuda().eval()
        B, T = 2, 4
        data = synth.make_batch(seed=9, B=B, T=T)
        data["query_idx"] = torch.tensor([3, 1])
        dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
        ref = model(dict(dev))
        logits_ref = model.decoder.last_logits.clone()
        cache = RefFeatureCache(model.rgb_encoder)
        cm = ref["camera_mask"]
        ref_imgs = dev["images"][~cm].reshape(B, T - 1, 3, 224, 224)
        feats = cache.encode(ref_imgs)
        d2 = dict(dev)
        d2["cached_rgb_feat"], d2["cached_rgb_mask"] = cache.place(feats, dev["query_idx"], T)
        out = model(d2)
        assert torch.equal(model.decoder.last_logits, logits_ref)
        assert torch.equal(out["pred_bbox"], ref["pred_bbox"])
        assert torch.equal(out["regression_boxes"], ref["regression_boxes"])
    # with a PROMOTED adapter the encoder hands its features over as split-f16 planes (another operand class in the cache): same bits
    # again, and features cached BEFORE the promotion state changed are refused loudly by the re-cast guard, never misread
    from boxdreamer_amd import _lib, calibrate
    model = BoxDreamer(_config("f16c8_qk16"))
    model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
    model = model.cuda().eval()
    st = calibrate.get_state(model.rgb_encoder, model.decoder)
    st["dec_misc"] = _lib.PROMOTE_ADAPTER_FC1 | _lib.PROMOTE_BBOX_PROJ
    st["enc"][1] = _lib.PROMOTE_PROJ | _lib.PROMOTE_FC2
    calibrate.set_state(model.rgb_encoder, model.decoder, st)
    model._calibrated_for = model.decoder._signature()                 # (this state is the test's: no self-calibration on top)
    ref = model(dict(dev))
    logits_ref = model.decoder.last_logits.clone()
    assert model.rgb_encoder.model.feats_class() == _lib.PREC_F16X3
    cache = RefFeatureCache(model.rgb_encoder)
    feats = cache.encode(dev["images"][~ref["camera_mask"]].reshape(B, T - 1, 3, 224, 224))
    d2 = dict(dev)
    d2["cached_rgb_feat"], d2["cached_rgb_mask"] = cache.place(feats, dev["query_idx"], T)
    model(d2)
    assert torch.equal(model.decoder.last_logits, logits_ref) and model.decoder.recast_count == 0
    # the documented flow encodes the references BEFORE the first forward -- i.e. before the load-time calibration may
    # have promoted ENCODER Linears while the adapter (and with it the features' operand class) stays as it was.  Such features are stale
    # in a way the operand class cannot show; the producer stamp on the tag does: the forward warns once and encodes every view afresh,
    # bit-identical to the uncached forward, instead of silently mixing two promotion states.
    import warnings
    from boxdreamer_amd import cache as cache_mod
    model = BoxDreamer(_config("f16c8_qk16"))
    model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
    model = model.cuda().eval()
    cache = RefFeatureCache(model.rgb_encoder)
    stale = cache.encode(dev["images"][~ref["camera_mask"]].reshape(B, T - 1, 3, 224, 224))      # un-promoted encoder
    st = calibrate.get_state(model.rgb_encoder, model.decoder)
    st["enc"][0] = _lib.PROMOTE_QKV | _lib.PROMOTE_FC1 | _lib.PROMOTE_FC2                          # what a calibration may find necessary
    calibrate.set_state(model.rgb_encoder, model.decoder, st)
    model._calibrated_for = model.decoder._signature()
    want = model(dict(dev))
    logits_want = model.decoder.last_logits.clone()
    d3 = dict(dev)
    d3["cached_rgb_feat"], d3["cached_rgb_mask"] = cache.place(stale, dev["query_idx"], T)
    cache_mod._WARNED_STALE = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = model(d3)
    assert any("encoding every view afresh" in str(x.message) for x in w)
    assert torch.equal(model.decoder.last_logits, logits_want) and torch.equal(got["pred_bbox"], want["pred_bbox"])
    fresh = cache.encode(dev["images"][~ref["camera_mask"]].reshape(B, T - 1, 3, 224, 224))      # re-encoded under the new state: cached path again
    d4 = dict(dev)
    d4["cached_rgb_feat"], d4["cached_rgb_mask"] = cache.place(fresh, dev["query_idx"], T)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model(d4)
    assert not any("encoding every view afresh" in str(x.message) for x in w)
    assert torch.equal(model.decoder.last_logits, logits_want)


def test_hip_graph_replay_matches_eager(hip):
    from boxdreamer_amd import hip_ops
    from boxdreamer_amd.graph import GraphedPath
    model = BoxDreamer(_config("bf16"))
    model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
    model = model.cuda().eval()
    B, T = 2, 3
    g = GraphedPath(model.rgb_encoder, model.decoder, B, T, 224, torch.float32, "cuda", want_idx=True)
    for seed, qi in ((8, [2, 0]), (9, [1, 1])):
        data = synth.make_batch(seed=seed, B=B, T=T)
        qidx = torch.tensor(qi)
        heat, kp, kn, idx = g(data["images"].cuda(), data["bbox_feat"].cuda(), qidx)
        mask = torch.zeros(B, T, dtype=torch.bool); mask[torch.arange(B), qidx] = True
        img, bf = data["images"].cuda(), data["bbox_feat"].cuda()
        ref = model.decoder(bf, img, mask.cuda(), model.rgb_encoder.predict(img), None)
        rkp, _, ridx = hip_ops.decode_topk(ref)
        assert torch.equal(heat, ref) and torch.equal(kp, rkp) and torch.equal(idx, ridx)
    # building a graph must not switch the decoder's one-hot mask check off for later eager use ...
    # (the facade's decoder leaves its verdict on the device -- "deferred": it travels with the corners' D2H, model.py; a stand-alone BETR checks
    # before the launch)
    assert model.decoder.validate_inputs == "deferred"
    model.decoder(bf, img, torch.zeros(B, T, dtype=torch.bool, device="cuda"), model.rgb_encoder.predict(img), None)
    assert bool(model.decoder.mask_error)
    model.decoder(bf, img, mask.cuda(), model.rgb_encoder.predict(img), None)
    assert not bool(model.decoder.mask_error)
    model.decoder.validate_inputs = True
    with pytest.raises(ValueError, match="exactly one query view"):
        model.decoder(bf, img, torch.zeros(B, T, dtype=torch.bool, device="cuda"), model.rgb_encoder.predict(img), None)
    model.decoder.validate_inputs = "deferred"
    # ... a live graph freezes the modules it captured raw pointers into (a larger eager batch would re-allocate the workspace) ...
    big = synth.make_batch(seed=3, B=B + 2, T=T)
    with pytest.raises(RuntimeError, match="GraphedPath"):
        model.rgb_encoder.predict(big["images"].cuda())
    # ... a SECOND graph on the same modules does not lift the first one's freeze when it is deleted, and deleting both does
    g2 = GraphedPath(model.rgb_encoder, model.decoder, B, T, 224, torch.float32, "cuda")
    del g2
    import gc; gc.collect()
    with pytest.raises(RuntimeError, match="GraphedPath"):
        model.rgb_encoder.predict(big["images"].cuda())
    del g
    gc.collect()
    assert model.rgb_encoder.predict(big["images"].cuda()).shape[0] == B + 2


@pytest.mark.parametrize("prec", ["f16c8_qk16", "bf16"])
def test_facade_hip_graph_and_sync_budget(hip, prec):
    """what a maintainer gets from the 3-line patch.  `hip_graph: true` replays the step from one captured graph behind
    BoxDreamer.forward -- every output bit-identical to the eager forward, over changing query positions and a changing batch shape -- and
    an eval forward waits for the device exactly once (the corners' D2H, which also carries the one-hot verdict of camera_mask)."""
    def build(graph):
        cfg = _config(prec)
        cfg["modules"]["hip_graph"] = graph
        m = BoxDreamer(cfg)
        m.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
        return m.cuda().eval()
    eager, graphed = build(False), build(True)
    for seed, B, T, qi in ((8, 2, 3, [2, 0]), (9, 2, 3, [1, 1]), (10, 3, 2, [0, 1, 1])):
        batch = synth.make_batch(seed=seed, B=B, T=T)
        batch["query_idx"] = torch.tensor(qi)
        dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
        a, b = eager(dict(dev)), graphed(dict(dev))
        for k in ("pred_bbox", "pred_poses", "regression_boxes", "pred_corners_px", "camera_mask"):
            assert torch.equal(a[k], b[k]), (prec, seed, k)
        assert len(eager.host_syncs_per_forward) == 1 and len(graphed.host_syncs_per_forward) == 1
        assert "ONE D2H" in graphed.host_syncs_per_forward[0]
    assert graphed._graph is not None and graphed._graph_key[0] == 3           # the last shape's capture replaced the first
    # a boolean-mask assignment or an .item() anywhere in the forward would show up here: torch's own count of synchronising calls
    import warnings
    for m in (eager, graphed):
        try:
            torch.cuda.set_sync_debug_mode("warn")
        except Exception:                       # noqa: BLE001 -- a build without the debug hook: the list above is the record
            break
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                m(dict(dev))
        finally:
            torch.cuda.set_sync_debug_mode("default")
        n = sum("synchroniz" in str(x.message).lower() for x in w)
        print(f"[facade syncs] {prec} {'graphed' if m is graphed else 'eager'}: torch counted {n} synchronising call(s) in one eval forward")
        assert n <= 1, [str(x.message) for x in w]


def test_two_batches_in_flight_match_one_at_a_time(hip):
    """bench.py's default step keeps two batches in flight: two independent captured copies of the path (own modules, workspace,
    graph) replayed alternately on two streams.  Every batch must still be computed in full: with different inputs on the two
    lanes, interleaved replays give, bit for bit, what each lane gives alone -- overlapping kernels of the other lane (they fill each
    other's idle CUs) must not leak into its buffers."""
    from boxdreamer_amd.graph import GraphedPath
    B, T = 2, 3
    lanes = []
    for seed in (21, 22):
        model = BoxDreamer(_config("bf16"))
        model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
        model = model.cuda().eval()
        g = GraphedPath(model.rgb_encoder, model.decoder, B, T, 224, torch.float32, "cuda")
        data = synth.make_batch(seed=seed, B=B, T=T)
        g.set_inputs(data["images"].cuda(), data["bbox_feat"].cuda(), torch.tensor([T - 1, 0]))
        lanes.append((g, torch.cuda.Stream()))
    torch.cuda.synchronize()
    alone = []
    for g, _ in lanes:                                   # one at a time, on the current stream
        heat, kp, _, _ = g.replay()
        torch.cuda.synchronize()
        alone.append((heat.clone(), kp.clone()))
    assert not torch.equal(alone[0][0], alone[1][0])      # the lanes really hold different batches
    for _ in range(3):                                   # interleaved, nothing between the two enqueues
        for g, s in lanes:
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        for (g, _), (heat0, kp0) in zip(lanes, alone):
            assert torch.equal(g.out[0], heat0) and torch.equal(g.out[1], kp0)


def _dense_model_and_batch(dense_cfg, B=2, T=7, seed=13, prec=STRICT_DEFAULT):
    cfg = _config(prec)
    cfg["modules"]["dense_cfg"] = dense_cfg
    model = BoxDreamer(cfg)
    model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
    model = model.cuda().eval()
    data = synth.make_batch(seed=seed, B=B, T=T)
    # crops in [0, 1] with a black border of a different width per view -> distinct foreground counts
    img = (data["images"].float() * 0.25 + 0.5).clamp(0, 1)
    for b in range(B):
        for t in range(T):
            w = 14 * ((3 * b + 2 * t) % 5)
            m = torch.zeros(224, 224); m[w:224 - w, w:224 - w] = 1
            img[b, t] *= m
    data["images"] = img
    data["query_idx"] = torch.tensor([T - 1, 1][:B])
    return model, data


@pytest.mark.parametrize("prec", [STRICT_DEFAULT, "bf16x3"])
def test_dense_mode_filter_path(hip, prec):
    """dense_cfg.enable + filter='dino' (BoxDreamerModel.py:291-330, data_processing.py:179-225): references are ranked
    by the HIP similarity kernel, the batch is re-packed (selected references in order, query last) and decoded once.
    The re-packed decode must equal the oracle run on the views the facade itself selected."""
    from boxdreamer_amd import dense
    from oracle import dense_oracle as do
    k = 3
    model, data = _dense_model_and_batch({"enable": True, "filter": "dino", "filter_enable": True, "filter_topk": k,
                                          "multi_round": False}, prec=prec)
    B, T = data["images"].shape[:2]
    dev = {kk: (v.cuda() if torch.is_tensor(v) else v) for kk, v in data.items()}
    out = model(dev)
    assert out["images"].shape[1] == k + 1 and out["bbox_feat"].shape[1] == k + 1 and out["poses"].shape[1] == k + 1
    assert out["query_idx"].tolist() == [k] * B and out["camera_mask"][:, -1].all()
    assert out["pred_bbox"].shape == (B, k + 1, 8, 224, 224)
    # which references were kept: recover from the re-packed images
    cm = torch.zeros(B, T, dtype=torch.bool); cm[torch.arange(B), data["query_idx"]] = True
    feats = model.rgb_encoder.predict(data["images"].cuda()).cpu()
    rf = feats[~cm].reshape(B, T - 1, *feats.shape[2:]); ri = data["images"][~cm].reshape(B, T - 1, 3, 224, 224)
    exact = do.dino_matching_scores_closed_form(rf, feats[cm], ri, data["images"][cm])
    sel = torch.zeros(B, T - 1, dtype=torch.bool)
    for b in range(B):
        for j in range(k):
            hit = [(ri[b, n] == out["images"][b, j].cpu()).all().item() for n in range(T - 1)]
            assert sum(hit) >= 1
            sel[b, hit.index(True)] = True
        assert exact[b][sel[b]].min().item() >= exact[b][~sel[b]].max().item() - 8e-3      # a valid top-k
    # decode parity on the selected views
    sub = {"images": do.filter_views(data["images"], cm, sel), "bbox_feat": do.filter_views(data["bbox_feat"].float(), cm, sel),
           "query_idx": torch.full((B,), k)}
    o = orc.boxdreamer_forward(sub, synth.betr_state_dict(1234, 2), synth.dino_state_dict(4321, 2))
    assert (model.decoder.last_logits.cpu() - o["logits"]).abs().max().item() <= 1e-3


def test_dense_mode_multi_round(hip):
    """multi_round: decoder over sub-batches of the references (+ query), one PnP over all rounds' corners, then either
    the coarse dict (fine_level False) or a fine decode on the pose-nearest references (dense_processing.py:8-158)."""
    base = {"enable": True, "filter": "dino", "filter_enable": False, "filter_topk": 4, "multi_round": True,
            "sub_batch_size": 3, "dense_mem_friendly": False, "fine_level": False, "fine_topk": 2}
    model, data = _dense_model_and_batch(base)
    B, T = data["images"].shape[:2]
    dev = {kk: (v.cuda() if torch.is_tensor(v) else v) for kk, v in data.items()}
    out = model(dev)
    assert out["pred_bbox"].shape == (B, T, 8, 224, 224)
    assert torch.isfinite(out["pred_poses"]).all()
    # VALUES of the coarse prediction: the query view's heatmaps are round 0's decode = the oracle on round 0's views
    # (references 0 .. sub-1 in order + the query last; oracle/dense_oracle.py pinned against the reference's sub_batchify)
    from oracle import dense_oracle as do
    cm = torch.zeros(B, T, dtype=torch.bool); cm[torch.arange(B), data["query_idx"]] = True
    ni = do.sub_batchify_views(data["images"], cm, base["sub_batch_size"])
    nb = do.sub_batchify_views(data["bbox_feat"].float(), cm, base["sub_batch_size"])
    o = orc.boxdreamer_forward({"images": ni[:, 0], "bbox_feat": nb[:, 0], "query_idx": torch.full((B,), base["sub_batch_size"])},
                               synth.betr_state_dict(1234, 2), synth.dino_state_dict(4321, 2))
    got = out["pred_bbox"].cpu()[cm].float()                              # (B, 8, 224, 224): 2 sigmoid(logits) - 1
    assert (got - o["heat"]).abs().max().item() <= 1e-3
    # ... and the query pose is the host PnP of ALL rounds' corners (checked for value against the solver on the oracle's corners
    # of round 0 only when there is a single round; here: finite, and a rigid transform)
    Rm = out["pred_poses"].cpu()[cm][:, :3, :3].double()
    assert (Rm @ Rm.transpose(-1, -2) - torch.eye(3, dtype=torch.float64)).abs().max().item() <= 1e-5
    assert out["pred_poses"].shape == (B, T, 4, 4)
    # both scheduling variants of the rounds give the same heatmaps
    model2, data2 = _dense_model_and_batch({**base, "dense_mem_friendly": True})
    out2 = model2({kk: (v.cuda() if torch.is_tensor(v) else v) for kk, v in data2.items()})
    assert torch.equal(out["pred_bbox"], out2["pred_bbox"])
    # fine level: a second decode on fine_topk references + query
    model3, data3 = _dense_model_and_batch({**base, "fine_level": True})
    out3 = model3({kk: (v.cuda() if torch.is_tensor(v) else v) for kk, v in data3.items()})
    assert out3["images"].shape[1] == 3 and out3["pred_bbox"].shape == (B, 3, 8, 224, 224)
    assert out3["camera_mask"][:, -1].all() and torch.isfinite(out3["pred_poses"]).all()
    # VALUES of the fine round (it was shape / finite-checked only): the re-packed batch holds fine_topk of the ORIGINAL
    # references (in their original order) + the query last, and the fine decode equals the oracle run on exactly those views
    for b in range(B):
        refs_b = [t for t in range(T) if not cm[b, t]]
        picked = []
        for j in range(2):
            hit = [t for t in refs_b if torch.equal(data3["images"][b, t], out3["images"][b, j].cpu())]
            assert len(hit) >= 1
            picked.append(hit[0])
            assert torch.equal(data3["bbox_feat"][b, hit[0]], out3["bbox_feat"][b, j].cpu())
        assert picked == sorted(picked) and len(set(picked)) == 2
        assert torch.equal(out3["images"][b, 2].cpu(), data3["images"][b][cm[b]][0])
    o3 = orc.boxdreamer_forward({"images": out3["images"].cpu().float(), "bbox_feat": out3["bbox_feat"].cpu().float(),
                                 "query_idx": torch.full((B,), 2)}, synth.betr_state_dict(1234, 2), synth.dino_state_dict(4321, 2))
    assert (model3.decoder.last_logits.cpu() - o3["logits"]).abs().max().item() <= 1e-3
    got3 = out3["pred_bbox"].cpu()[out3["camera_mask"].cpu()].float()
    assert (got3 - o3["heat"]).abs().max().item() <= 1e-3
    assert (out3["pred_corners_px"].cpu() - o3["corners_px"]).abs().max().item() <= 224 / 20 * 2


# ---------------------------------------------------------------- pose VALUES through process_prediction

def _posed_batch(B, T, seed=3):
    """A geometrically consistent batch: per sample a box (bbox_3d), intrinsics and a known query pose; the query view's
    TRUE corner projections are returned so that a stub decoder can emit heatmaps peaked exactly there."""
    import numpy as np
    from boxdreamer_amd import pnp
    rng = np.random.default_rng(seed)
    data = synth.make_batch(seed=seed, B=B, T=T)
    f = 1.2 * 224
    K = np.array([[f, 0, 112.0], [0, f, 112.0], [0, 0, 1.0]])
    box = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float) * [0.45, 0.35, 0.3]
    poses = np.tile(np.eye(4), (B, T, 1, 1))
    proj = np.zeros((B, 8, 2))
    for b in range(B):
        R = pnp.rodrigues(rng.normal(size=3) * 0.6)
        t = np.array([rng.normal() * 0.1, rng.normal() * 0.1, 2.6 + rng.random() * 0.6])
        pc = box @ R.T + t
        proj[b] = pc[:, :2] / pc[:, 2:3] * f + 112.0
        poses[b, T - 1, :3, :3], poses[b, T - 1, :3, 3] = R, t
    data["bbox_3d"] = torch.from_numpy(np.tile(box, (B, T, 1, 1))).float()
    data["non_ndc_intrinsics"] = torch.from_numpy(np.tile(K, (B, T, 1, 1))).float()
    data["intrinsics"] = data["non_ndc_intrinsics"].clone()
    data["poses"] = torch.from_numpy(poses).float()
    return data, torch.from_numpy(proj).float()


@pytest.mark.parametrize("on_device", [False, True])
def test_pose_values_from_exact_corner_heatmaps(hip, on_device):
    """a8 (prediction_utils.py:63-101, box_utils.py:113-199) with VALUES: the decoder is replaced by a stub that emits the
    dataset-style corner heatmaps (make_bbox_features, rendered by bd_render_corner_heatmaps) of the query view's TRUE
    projected corners; the facade's decode (top-20 mean) + PnP must then return the known query pose.  Expected accuracy
    follows from the decode: the top-20 mean sits within ~0.5 px of the true corner, boxes span ~90 px at f = 269 px and
    z ~ 2.9, so R is recovered to ~0.02 and t to ~2 % of the depth.  The pose matrix itself (OpenCV's solvePnP) stays
    un-pinned in this image (no cv2): `pose_solver` says which solver ran."""
    from boxdreamer_amd.bbox_features import make_bbox_features
    cfg = _config("bf16")
    cfg["modules"]["pnp_on_device"] = on_device
    model = BoxDreamer(cfg).cuda().eval()
    B, T = 5, 3
    data, proj = _posed_batch(B, T)
    gt = data["poses"][:, T - 1].clone()
    data["poses"][:, T - 1] = torch.eye(4)                       # the query pose is what must be recovered
    heat = make_bbox_features(proj.cuda(), "heatmap", (224, 224), group=1)   # (B, 8, 224, 224) in [-1, 1], peaks at the projections

    class Stub(torch.nn.Module):
        def forward(self, *a, **k):
            return heat.float()
    model.decoder = Stub()
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    out = model(dev)
    kp = out["pred_corners_px"].cpu()
    assert (kp - proj).abs().max().item() <= 0.75               # top-20 mean of a peaked map: sub-pixel
    pp = out["pred_poses"][:, T - 1].float().cpu()
    assert ("hip:bd_solve_pnp" in out["pose_solver"]) == on_device and "un-pinned" in out["pose_solver"] or "cv2" in out["pose_solver"]
    assert (pp[:, :3, :3] - gt[:, :3, :3]).abs().max().item() <= 0.03
    assert ((pp[:, :3, 3] - gt[:, :3, 3]).abs() / gt[:, 2:3, 3]).max().item() <= 0.03
    assert torch.equal(pp[:, 3], torch.tensor([0.0, 0, 0, 1]).expand(B, 4))
    assert torch.equal(out["pred_poses"][:, : T - 1].cpu(), data["poses"][:, : T - 1])


def test_configs2_substitute_b64_end_to_end(hip):
    """BASELINE configs[2] (LINEMOD eval, pretrained checkpoint, 5 references, batch 64, CPU PnP) needs files that are not
    available offline (checkpoint: run.py:172-183; data: configs/test.yaml:18-24).  SURVEY §8d's substitute: synthetic
    B = 64, T = 6 at FULL depth through the facade -> heatmaps -> corners -> host PnP, in the strict mode, with the oracle
    on two of the 64 samples and batch-independence (bit-exact) on two more.  Runs in the facade's DEFAULT mode."""
    cfg = _config(None, depth=12)                   # the facade's default mode (f16c8_qk16)
    model = BoxDreamer(cfg)
    assert model.decoder.hip_precision == STRICT_DEFAULT
    model.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 12).items()}, strict=True)
    model = model.cuda().eval()
    B, T = 64, 6
    small = synth.make_batch(seed=61, B=8, T=T, dtype=torch.bfloat16)
    data = {k: (v.repeat(8, *([1] * (v.dim() - 1))) if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 8 else v)
            for k, v in small.items()}
    for r in range(8):                                            # 64 distinct samples (offsets exact in bf16)
        data["images"][8 * r:8 * r + 8] += 0.03125 * r
    data["query_idx"] = torch.arange(B) % T
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    out = model(dev)
    logits = model.decoder.last_logits.clone()
    assert out["pred_bbox"].shape == (B, T, 8, 224, 224) and out["pred_bbox"].dtype == torch.bfloat16
    assert out["pred_poses"].shape == (B, T, 4, 4) and torch.isfinite(out["pred_poses"].float()).all()
    assert out["regression_boxes"].shape == (B, T, 8, 2) and out["pose_solver"].startswith("host:")
    cm = out["camera_mask"]
    assert cm.sum(1).tolist() == [1] * B and torch.equal(cm.float().argmax(1).cpu(), data["query_idx"])
    assert torch.equal(out["pred_poses"][~cm], dev["poses"][~cm])
    kp = out["pred_corners_px"]
    assert kp.shape == (B, 8, 2) and (kp >= 0).all() and (kp <= 223).all()
    for b in (0, 37):                                             # oracle on single samples (full depth: ~1 s each)
        one = {k: (v[b:b + 1].float() if torch.is_tensor(v) and v.is_floating_point() and v.shape[0] == B else
                   (v[b:b + 1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v)) for k, v in data.items()}
        o = orc.boxdreamer_forward(one, synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12))
        assert (logits[b].cpu() - o["logits"][0]).abs().max().item() <= 1e-3
        assert (kp[b].cpu() - o["corners_px"][0]).abs().max().item() <= 224 / 20 * 2
    for b in (5, 63):                                             # sample b alone == sample b inside the batch of 64
        single = {k: (v[b:b + 1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in dev.items()}
        model(single)
        assert torch.equal(model.decoder.last_logits[0], logits[b])


# ---------------------------------------------------------------- precision self-check at the drop-in boundary

def _facade_with(bsd, dsd, depth, **mods):
    cfg = _config(None, depth)
    cfg["modules"]["encoder"]["dino"]["cfg"] = {"model_type": "dinov2_vitb14_reg", "freeze": True, "state_dict": dsd}
    cfg["modules"].update(mods)
    model = BoxDreamer(cfg)
    model.load_state_dict({"decoder." + k: v for k, v in bsd.items()}, strict=True)
    return model.cuda().eval()


def test_facade_self_check_and_promotion_on_trained_like_weights(hip):
    """A maintainer who loads a checkpoint with outlier channels gets (1) a warning that the default mode needed promotion, (2) the
    promoted mode inside 1e-3 ABSOLUTE of the fp32 forward, (3) the whole record in the output dict -- without setting anything."""
    import warnings
    depth, gain = 12, 0.5
    bsd, dsd = synth.betr_state_dict_outliers(1234, depth, gain), synth.dino_state_dict_outliers(4321, depth, gain)
    data = synth.make_batch(seed=11, B=2, T=2)
    o = orc.boxdreamer_forward(data, bsd, dsd)
    dev = lambda: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    model = _facade_with(bsd, dsd, depth)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = model(dev())
    rec = out["hip_precision"]
    print("[facade self-check] " + str(rec))
    assert any("promoted to split-f16" in str(x.message) for x in w)
    assert rec["decoder"] == rec["encoder"] == STRICT_DEFAULT and rec["source"] == "package default"
    assert rec["calibrated"] and rec["promoted_units"] > 0 and rec["self_check_ok"] and rec["self_check_max_abs_dlogits"] <= 4e-4
    assert rec["self_check_unpromoted"] > 4e-4
    e = (model.decoder.last_logits.cpu() - o["logits"]).abs().max().item()
    print(f"[facade self-check] logits vs the CPU oracle after promotion: {e:.3e}")
    assert e <= 1e-3
    # the second forward does not calibrate again (same weights): no warning, same bits
    lg = model.decoder.last_logits.clone()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model(dev())
    assert torch.equal(model.decoder.last_logits, lg)
    # measuring only: the check fails LOUDLY and says so in the dict
    model2 = _facade_with(bsd, dsd, depth, hip_calibrate=False)
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        out2 = model2(dev())
    assert any("promotion is disabled" in str(x.message) for x in w2)
    assert out2["hip_precision"]["self_check_ok"] is False and out2["hip_precision"]["promoted_units"] == 0
    # a promotion state the caller applied is neither measured over by the first forward nor wiped by a measuring-only
    # calibration -- the measured state of `model` moved to a fresh facade keeps its bits and its error
    from boxdreamer_amd import calibrate
    good = calibrate.get_state(model.rgb_encoder, model.decoder)
    model3 = _facade_with(bsd, dsd, depth)
    calibrate.set_state(model3.rgb_encoder, model3.decoder, good)
    with warnings.catch_warnings():
        warnings.simplefilter("error")              # no calibration, hence no "promoted N units" warning
        model3(dev())
    assert calibrate.get_state(model3.rgb_encoder, model3.decoder) == good and torch.equal(model3.decoder.last_logits, lg)
    model4 = _facade_with(bsd, dsd, depth, hip_calibrate=False)
    calibrate.set_state(model4.rgb_encoder, model4.decoder, good)
    model4._calibrated_for = None
    rep4 = model4.calibrate(dev())                   # measuring only: reports the applied state's error and leaves it in place
    assert calibrate.get_state(model4.rgb_encoder, model4.decoder) == good
    assert rep4["delta_unpromoted"] > 4e-4 and rep4["delta_entry_state"] <= 4e-4 and rep4["ok"] and len(rep4["promoted"]) == rec["promoted_units"]
    # an explicit mode outside the F16C8 family is recorded as such and not touched
    cfg = _config("bf16", 2)
    m3 = BoxDreamer(cfg).cuda().eval()
    m3.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
    r3 = m3(dev())["hip_precision"]
    assert r3["decoder"] == "bf16" and r3["source"] == "config" and r3["calibrated"] is False


def test_real_checkpoint_parity_when_files_are_given(hip):
    """configs[2] (LINEMOD eval, /root/reference/configs/test.yaml:18-24, checkpoint loading run.py:172-183) needs files this image
    cannot fetch.  OPT-IN: with $BOXDREAMER_CKPT (BoxDreamer-vitb.safetensor / a Lightning .ckpt) and $BOXDREAMER_DINO_WEIGHTS
    (dinov2_vitb14_reg4 state_dict) pointing at real files, the default facade -- self-check and promotion included -- must match the
    fp32 CPU oracle on those weights within 1e-3 on the heatmap logits, with identical top-20 sets."""
    import os
    ckpt, dino = os.environ.get("BOXDREAMER_CKPT"), os.environ.get("BOXDREAMER_DINO_WEIGHTS")
    if not (ckpt and dino and os.path.isfile(ckpt) and os.path.isfile(dino)):
        pytest.skip("set $BOXDREAMER_CKPT and $BOXDREAMER_DINO_WEIGHTS to real checkpoint files to pin configs[2]")
    from boxdreamer_amd.encoder import _load_state_dict_file
    raw = _load_state_dict_file(ckpt)
    strip = lambda k: k[len("BoxDreamer."):] if k.startswith("BoxDreamer.") else k          # demo.py:564-573
    bsd = {strip(k)[len("decoder."):]: v.float() for k, v in raw.items() if strip(k).startswith("decoder.")}
    dsd = {k: v.float() for k, v in _load_state_dict_file(dino).items()}
    depth = 1 + max(int(k.split(".")[1]) for k in bsd if k.startswith("attn."))
    model = _facade_with(bsd, dsd, depth)
    data = synth.make_batch(seed=21, B=2, T=6)
    out = model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()})
    o = orc.boxdreamer_forward(data, bsd, dsd)
    e = (model.decoder.last_logits.cpu() - o["logits"]).abs().max().item()
    print(f"[real checkpoint] logits max-abs err {e:.3e}; " + str(out["hip_precision"]))
    assert e <= 1e-3
    assert (out["pred_corners_px"].cpu() - o["corners_px"]).abs().max().item() <= 1e-3
