"""Where does an eval forward of the facade wait for the device?  torch's sync debug mode ("warn") with a stack per warning."""
import copy, json, os, sys, traceback, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxdreamer_amd import synth
from boxdreamer_amd.model import BoxDreamer

mods = copy.deepcopy(json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model_modules_config.json")))["modules"])
mods["decoder"].update(num_decoder_layers=2, hip_precision="f16c8_qk16")
mods["encoder"]["dino"]["cfg"].update(synthetic_seed=4321, depth=2, hip_precision="f16c8_qk16")
mods["hip_graph"] = "--graph" in sys.argv
m = BoxDreamer({"modules": mods})
m.load_state_dict({"decoder." + k: v for k, v in synth.betr_state_dict(1234, 2).items()}, strict=True)
m = m.cuda().eval()
b = synth.make_batch(seed=8, B=2, T=3)
dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
m(dict(dev)); m(dict(dev))
torch.cuda.synchronize()

def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message).lower():
        st = [f for f in traceback.extract_stack() if "boxdreamer_amd" in f.filename]
        print("SYNC:", " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[::-1][:4]))
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
m(dict(dev))
torch.cuda.set_sync_debug_mode("default")
print(m.host_syncs_per_forward)
