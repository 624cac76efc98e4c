"""Kernel time of the BASELINE configs (and nefertiti) at full size under values of one rtc_set_option:
python tools/option_time_full.py claim_ahead=-1 claim_ahead=4 ...   (GPU box)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
CASES = [("cover.json", 1920, 1080, 5), ("reflection_and_refraction.json", 1920, 1080, 8), ("teapot.json", 1920, 1080, 5),
         ("nefertiti.json", 1920, 1080, 5), ("dragons.json", 3840, 2160, 5), ("fresnel.json", 300, 300, 5)]
scenes = {}
def t(name, w, h, depth):
    if name not in scenes: scenes[name] = rtc.HostScene.from_file(name)
    hs = scenes[name]; cam = hs.camera(w, h)
    canvas = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    ts = []
    for rep in range(3):
        gpu = rtc.GpuScene(hs.desc)
        for i in range(50):  # (the schedule settles over the first frames)
            gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(20): gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
        b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20); gpu.close()
    return min(ts)
for opt in sys.argv[1:]:
    n, v = opt.split("="); rtc.set_option(n, float(v))
    print(opt, " ".join(f"{c[0].split('.')[0]} {t(*c):.4f}" for c in CASES), flush=True)
