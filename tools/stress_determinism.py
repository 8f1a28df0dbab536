"""Race screen (python tools/stress_determinism.py <prec> <runs> [batch] [lanes] [latency]): the full-depth step (T=6; default B=32) repeated N times must give bit-identical logits
every time (LDS-DMA / barrier schedules: a RAW race shows up as rare differing tiles).  The reference run is the ONE-LANE step; the repeated runs use `lanes`
("auto" by default: two sub-batch lanes at B >= 11), so a cross-lane overlap of workspace slices or a missing join shows up here too."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import synth
from boxdreamer_amd.betr import BETR
from boxdreamer_amd.encoder import DinoV2Wrapper
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32        # (B < 32: the small-batch forms of the GEMMs, e.g. the 4-stage one-tile ring)
enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "synthetic_seed": 4321, "depth": 12, "hip_precision": prec})
dec = BETR(d_model=768, nhead=8, num_decoder_layers=12, decoder_only=True, patch_size=14, img_size=224, diff_emb=False,
           nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True, patchify_rays=True,
           pose_representation="bb8", bbox_representation="heatmap", hip_precision=prec)
dec.load_state_dict(synth.betr_state_dict(1234, 12), strict=True); dec = dec.cuda().eval()
small = synth.make_batch(seed=41, B=min(B, 4), T=6)
rep = max(1, B // 4)
img = small["images"].repeat(rep, 1, 1, 1, 1).to(torch.bfloat16).cuda()
bf = small["bbox_feat"].repeat(rep, 1, 1, 1, 1).to(torch.bfloat16).cuda()
mask = torch.zeros(img.shape[0], 6, dtype=torch.bool, device="cuda"); mask[:, 5] = True
lanes = sys.argv[4] if len(sys.argv) > 4 else "auto"
lanes = lanes if lanes == "auto" else int(lanes)
if len(sys.argv) > 5 and sys.argv[5] == "latency":      # the opt-in latency forms (split-K hand-over through flags: a lost / early flag would show up as differing tiles)
    enc.model.latency, dec.hip_latency = True, True
enc.model.lanes, dec.hip_lanes = 1, 1
dec(bf, img, mask, enc.predict(img), None)
ref = dec.last_logits.clone()
enc.model.lanes, dec.hip_lanes = lanes, lanes
bad = 0
for i in range(n):
    dec(bf, img, mask, enc.predict(img), None)
    l = dec.last_logits.clone()
    if not torch.equal(ref, l):
        bad += 1
        print("run", i, "differs: max", (ref - l).abs().max().item(), "count", int((ref != l).sum()))
print(f"{prec} B={img.shape[0]} lanes={lanes}{' latency forms' if dec.hip_latency else ''}: {n} runs against the one-lane step, {bad} differing")
