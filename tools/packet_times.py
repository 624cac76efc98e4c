import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
rtc = importlib.import_module("ray-tracer-challenge_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cover.json"
hs = rtc.HostScene.from_file(name); gpu = rtc.GpuScene(hs.desc); cam = hs.camera(1920, 1080)
depth = 8 if name.startswith("reflection") else 5
canvas = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for _ in range(6): gpu.render_device(cam, canvas.data_ptr(), depth, None, stream.cuda_stream)
torch.cuda.synchronize()
print(gpu.stats())
