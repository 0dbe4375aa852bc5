import sys, time, torch
sys.path.insert(0, '.')
from benchmarks import workloads as bw
from pdanet_amd import pointnet2_batch_cuda as ext
dev = torch.device('cuda:0')
ext.fps_coop_timeouts(reset=True)
for name, batch, pts, N in (('detector_train', 2, 60000, 400), ('detector_train', 8, 65536, 60), ('backbone_infer', 2, 60000, 200)):
    wl = bw.create(name, batch, pts, dev, 0, 1); wl.begin()
    losses = []
    t = time.perf_counter()
    for i in range(N):
        l = wl.step()
        if i % (N // 5) == 0 or i == N - 1:
            losses.append(float(l.detach().float().mean()))
    torch.cuda.synchronize()
    ok = all(x == x and abs(x) < 1e6 for x in losses)
    print(name, batch, pts, "%d steps in %.1f s" % (N, time.perf_counter() - t), "finite" if ok else "NON-FINITE", " ".join("%.3f" % x for x in losses),
          "coop FPS timeouts", ext.fps_coop_timeouts(), flush=True)
    del wl
