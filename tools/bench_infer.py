"""Full-image inference rate: render_path over synthetic poses at the LLFF evaluation
size (378 x 504, 64 + 128 samples), forward only.  Prints one JSON line (not the driver's bench)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(images=6, chunk=32768, height=378, width=504):
    import types
    a = types.SimpleNamespace(images=images, chunk=chunk, height=height, width=width)
    from scnerf_amd import create_nerf as cn, render as R, run_nerf_helpers as h, synthetic as synth
    H, W = a.height, a.width

    def net(seed):
        m = h.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict(synth.network_params(seed=seed))
        return m.cuda()
    kw = dict(network_query_fn=cn.FusedNetworkQuery(h.get_embedder(10, 0)[0], h.get_embedder(4, 0)[0]), perturb=0.0,
              N_importance=128, network_fine=net(1), N_samples=64, network_fn=net(0), use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0.0, near=0., far=1., ndc=True)
    poses = synth.camera_spec(H, W, n_cams=a.images, seed=4)["poses"]
    K = torch.tensor([[400.0, 0, W / 2, 0], [0, 400.0, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]]).cuda()
    common = dict(gt_intrinsic=K, gt_extrinsic=poses.cuda())
    R.render_path(poses[:1], (H, W, None), a.chunk, kw, "test", **common)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rgbs, _ = R.render_path(poses, (H, W, None), a.chunk, kw, "test", **common)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = a.images * H * W
    flop = rays * (64 * 2 + 128) * 2 * 593408            # coarse 64 + fine 192 samples, fwd MACs/sample x2
    return {"metric": "rays/sec (64+128 samples/ray) full-image inference", "value": rays / dt,
            "unit": "rays/s", "images": a.images, "image": [H, W], "chunk": a.chunk,
            "s_per_image": dt / a.images, "tflops_algorithmic": flop / dt / 1e12,
            "includes": "ray generation, NDC, render, D2H copy into numpy"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=6)
    ap.add_argument("--chunk", type=int, default=32768)
    ap.add_argument("--height", type=int, default=378)
    ap.add_argument("--width", type=int, default=504)
    a = ap.parse_args()
    print(json.dumps(run(a.images, a.chunk, a.height, a.width)))


if __name__ == "__main__":
    main()
