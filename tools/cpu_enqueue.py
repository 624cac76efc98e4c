import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
rtc = importlib.import_module("ray-tracer-challenge_amd")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
hs = rtc.HostScene.from_file("cover.json"); cam = hs.camera(1920, 1080); gpu = rtc.GpuScene(hs.desc)
T = 64; tx, ty = rtc.tile_grid(1920, 1080, T, T); first, stride, count, padded = rtc.tiles_of_rank(tx*ty, 0, 1)
buf = torch.zeros((padded, T, T, 3), dtype=torch.float64, device="cuda")
gathered = torch.empty((1,) + tuple(buf.shape), dtype=torch.float64, device="cuda")
canvas = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); comm = torch.cuda.Stream()
ev = torch.cuda.Event()
def step():
    gpu.render_tiles_device(cam, buf.data_ptr(), T, T, first, stride, count, 5, stream.cuda_stream)
    ev.record(stream)
    with torch.cuda.stream(comm):
        comm.wait_event(ev)
        dist.gather(buf, [gathered[0]], dst=0)
        rtc.assemble_tiles_device(gathered.data_ptr(), 1, padded, T, T, 1920, 1080, canvas.data_ptr(), comm.cuda_stream)
for _ in range(5): step()
torch.cuda.synchronize()
for part in ("render", "gather", "assemble", "all"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        if part == "render": gpu.render_tiles_device(cam, buf.data_ptr(), T, T, first, stride, count, 0, stream.cuda_stream)
        elif part == "gather":
            dist.gather(buf, [gathered[0]], dst=0)
        elif part == "assemble": rtc.assemble_tiles_device(gathered.data_ptr(), 1, padded, T, T, 1920, 1080, canvas.data_ptr(), comm.cuda_stream)
        else: step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(part, "cpu enqueue us/step", round((t1 - t0) / 200 * 1e6, 1), "incl sync us/step", round((t2 - t0) / 200 * 1e6, 1))
dist.destroy_process_group()
