"""Experiment: do two independent lock-step batches on two HIP streams (two executors, two host threads) fill the tails
that one stream leaves?  Prints images/s for one stream with n images and for two streams with n images each.
Usage: python tools/two_stream.py [n] [diffusion_steps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit.engine import HEditEngine
from hedit.p2p import ptp_controller_utils as PCU
from hedit.p2p.ptp_classes import ControllerBatch
from hedit.p2p.ptp_utils import register_attention_control
from hedit.pipeline import HEditPipeline
from hedit.scheduler import DDIMScheduler
from hedit.text import ClipTextEncoder, WordTokenizer
from hedit.unet import SD15_CONFIG, UNet2DConditionModel, random_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
PAIRS = [("a cat sitting on a bench", "a dog sitting on a bench", ("cat", "dog"), True),
         ("a photo of a red car", "a photo of a blue car", ("red", "blue"), True)]
sd = random_state_dict(UNet2DConditionModel(SD15_CONFIG, device=dev).param_shapes, seed=0)


def make(seed):
    unet = UNet2DConditionModel(SD15_CONFIG, device=dev)
    unet.load_state_dict(sd)
    tok = WordTokenizer()
    enc = ClipTextEncoder(dim=768, layers=12, heads=12, seed=7).to(dev)
    model = HEditPipeline(unet, DDIMScheduler(), tok, enc, None, dev)
    model.scheduler.set_timesteps(T)
    eng = HEditEngine(model)
    pairs = [PAIRS[i % 2] for i in range(n)]
    pp = [[p[0], p[1]] for p in pairs]
    w0 = torch.randn(n, 4, 64, 64, generator=torch.Generator().manual_seed(seed)).to(dev) * 0.8
    null = eng.encode([""]); src = eng.encode([p[0] for p in pp]); tar = eng.encode([p[1] for p in pp])
    zs, xts = eng.ddpm_inversion(w0, [p[0] for p in pp], eta=1.0, cfg_src=1.0, generator=torch.Generator(device=dev).manual_seed(seed))
    xT = xts[T].contiguous()

    def step():
        cb = ControllerBatch([PCU.make_controller(prompts=[s_, t_], is_replace_controller=r, cross_replace_steps=0.4, self_replace_steps=0.35,
                                                  blend_word=((bw[0],), (bw[1],)), equilizer_params={"words": (bw[1],), "values": (2.0,)},
                                                  num_steps=T, tokenizer=tok, device=dev) for (s_, t_, bw, r) in pairs])
        register_attention_control(model, cb)
        return eng.run(xT, zs, pp, [1.0, 5.0, 7.5], cb, eta=1.0, p2p=True, implicit=True, K=1, w_rec=0.1, after_skip_steps=T, ddim_inv=False,
                       ctx=(null, src, tar), fuse_src_pass=True)
    return step


a, b = make(1), make(2)
ra = a(); torch.cuda.synchronize()
t0 = time.perf_counter(); ra2 = a(); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"one stream, {n} images, {T} steps: {t1:.3f} s -> {n / t1 * T / 50:.3f} images/s (50-step equivalent)", flush=True)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [None, None]


def worker(i, fn):
    with torch.cuda.stream(streams[i]):
        outs[i] = fn()


for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i, f)) for i, f in enumerate((a, b))]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"two streams, {n} images each: {t2:.3f} s -> {2 * n / t2 * T / 50:.3f} images/s; stream-0 result identical to the solo run: {torch.equal(outs[0][0], ra2[0])}", flush=True)
