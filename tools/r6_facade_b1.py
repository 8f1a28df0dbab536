"""Round 6: one pose per call THROUGH THE FACADE (the reference demo's loop, src/demo/demo.py:1501-1514): BoxDreamer(config).eval()(batch) at
B = 1, T = 6 with hip_graph, default forms against `hip_latency: true`; ms per call incl. the D2H, the host PnP and the dict's outputs."""
import copy, json, os, sys, time, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from boxdreamer_amd import synth
from boxdreamer_amd.model import BoxDreamer

prec, dev = "f16c8_qk16", torch.device("cuda")
bsd, dsd = bench.state_dicts("plain")
one = synth.make_batch(seed=100, B=1, T=6)
batch = {k: ((v.to(torch.bfloat16) if v.is_floating_point() else v).to(dev) if torch.is_tensor(v) else v) for k, v in one.items()}
for lat in (False, True):
    for graph in (True, False):
        mods = copy.deepcopy(json.load(open(os.path.join(ROOT, "tests", "golden", "model_modules_config.json")))["modules"])
        mods["decoder"].update(num_decoder_layers=12, hip_precision=prec)
        mods["encoder"]["dino"]["cfg"].update(state_dict=dsd, hip_precision=prec)
        mods["hip_graph"], mods["hip_latency"] = graph, lat
        m = BoxDreamer({"modules": mods})
        m.load_state_dict({"decoder." + k: v for k, v in bsd.items()}, strict=True)
        m = m.to(dev).eval()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(5):
                out = m(dict(batch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            out = m(dict(batch))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 100 * 1e3
        print(f"facade B=1 hip_latency={lat} hip_graph={graph}: {ms:.3f} ms per call ({1e3 / ms:.1f} frames/s); pose finite {bool(torch.isfinite(out['pred_poses']).all())}", flush=True)
        del m
        torch.cuda.empty_cache()
