"""Round 6: where the facade's forward (hip_graph mode, configs[1] dict, default precision) spends HOST time -- cProfile over 20 forwards,
plus the wall time per forward split at the D2H: [enqueue until the blocking copy | wait | PnP + tail]."""
import cProfile, copy, io, json, os, pstats, sys, time, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from boxdreamer_amd import synth
from boxdreamer_amd.model import BoxDreamer

prec, dev = "f16c8_qk16", torch.device("cuda")
mods = copy.deepcopy(json.load(open(os.path.join(ROOT, "tests", "golden", "model_modules_config.json")))["modules"])
bsd, dsd = bench.state_dicts("plain")
mods["decoder"].update(num_decoder_layers=12, hip_precision=prec)
mods["encoder"]["dino"]["cfg"].update(state_dict=dsd, hip_precision=prec)
mods["hip_graph"] = True
m = BoxDreamer({"modules": mods})
m.load_state_dict({"decoder." + k: v for k, v in bsd.items()}, strict=True)
m = m.to(dev).eval()
one = synth.make_batch(seed=100, B=32, T=6)
batch = {k: ((v.to(torch.bfloat16) if v.is_floating_point() else v).to(dev) if torch.is_tensor(v) else v) for k, v in one.items()}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(4):
        m(dict(batch))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    m(dict(batch))
torch.cuda.synchronize()
print("ms per forward", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    m(dict(batch))
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
