"""Per-kernel timings on the GPU box (HIP events on the launch stream).  Scratch tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as Hh


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def preproc():
    from vipformer_amd.model.pointcloud import utils as U
    for (B, N, G, K) in [(128, 1024, 96, 32), (64, 1024, 128, 32), (32, 2048, 128, 32)]:
        pts = Hh.synth_points(1, B, N).cuda(); start = Hh.synth_start(1, B, N).cuda()
        t_fps = timeit(lambda: U._fps_from_start(pts, G, start))
        idx = U._fps_from_start(pts, G, start); ct = U.index_points(pts, idx)
        t_knn = timeit(lambda: U._knn_group(pts, ct, K, True, False, False, True))
        fps_bytes = B * (12 * N + 8 * G); knn_bytes = B * (12 * N + 12 * G + 12 * G * K + 12 * G)
        print(f"preproc B={B} N={N} G={G}: fps {t_fps:.1f} us ({t_fps/G*1e3:.0f} ns/iter, {fps_bytes/t_fps/1e3:.2f} GB/s)  "
              f"knn+group {t_knn:.1f} us ({knn_bytes/t_knn/1e3:.2f} GB/s)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["preproc"]
    for w in which:
        globals()[w]()
