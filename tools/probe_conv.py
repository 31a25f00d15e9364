"""GPU probe: MIOpen fp32 conv throughput of the BEV backbone + tail in NCHW vs channels_last.
Design input only (decides the feature-map layout the HIP kernels target); not part of the product path."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd.backbone import ResNetBEVBackbone, DownsampleConv

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def main():
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0), torch.__version__)
    cfg = dict(layer_nums=[3,5,8], layer_strides=[2,2,2], num_filters=[64,128,256],
               upsample_strides=[1,2,4], num_upsample_filter=[128,128,128])
    torch.manual_seed(0)
    bb = ResNetBEVBackbone(cfg, 64).to(dev).eval()
    shrink = DownsampleConv(dict(kernal_size=[3], stride=[1], padding=[1], dim=[256], input_dim=384)).to(dev).eval()
    res = {}
    # HBM copy ceiling
    a = torch.empty(256*1024*1024, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a)); res["copy_GBps"] = 2*a.numel()*4/t/1e6
    t = timeit(lambda: a.zero_()); res["memset_GBps"] = a.numel()*4/t/1e6
    del a, b
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
            bbf = bb.to(memory_format=fmt); shf = shrink.to(memory_format=fmt)
            for N in (1, 5):
                x = torch.randn(N, 64, 200, 704, device=dev).contiguous(memory_format=fmt)
                with torch.no_grad():
                    t_bb = timeit(lambda: bbf.get_multiscale_feature(x))
                    feats = bbf.get_multiscale_feature(x)
                    ego = [f[:1].contiguous(memory_format=fmt) for f in feats]
                    t_tail = timeit(lambda: shf(bbf.decode_multiscale_feature(ego)))
                res[f"bench{int(bench)}_{fmt_name}_N{N}"] = dict(backbone_ms=t_bb, tail_ms=t_tail,
                    backbone_TFLOPs=81e9*N/t_bb/1e9, tail_TFLOPs=108e9/t_tail/1e9)
                print(f"bench={bench} {fmt_name} N={N}: backbone {t_bb:.3f} ms  tail {t_tail:.3f} ms", flush=True)
    print(json.dumps(res, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/probe_conv.json", "w"), indent=1)

if __name__ == "__main__":
    main()
