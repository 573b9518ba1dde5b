"""Micro-timing of the policy step pieces at B=4096 (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from envs import spaces
from models import MultigridNetwork
dev = "cuda:0"; B = 4096
obs_space = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
net = MultigridNetwork(obs_space, spaces.Discrete(7), scalar_fc=5, scalar_dim=4).to(dev)
img = torch.rand(B, 3, 5, 5, device=dev); d = torch.randint(0, 4, (B, 1), device=dev).float()
h = (torch.zeros(B, 256, device=dev), torch.zeros(B, 256, device=dev)); m = torch.ones(B, 1, device=dev)
def t(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter()-t0)/n*1e6:9.1f} us")
with torch.no_grad():
    t("conv (dispatch)", lambda: net.image_conv(img))
    t("conv gemm explicit", lambda: net.image_conv[0]._valid_conv_gemm(img))
    big = torch.rand(262144, 3, 5, 5, device=dev)
    t("conv gemm 262144 imgs", lambda: net.image_conv[0]._valid_conv_gemm(big), n=5)
    t("conv miopen", lambda: F.conv2d(img, net.image_conv[0].weight, net.image_conv[0].bias))
    t("unfold only", lambda: F.unfold(img, (3, 3)))
    x = torch.rand(B, 149, device=dev)
    t("lstm manual step", lambda: net.rnn._lstm_cuda(x, h, m, True))
    t("lstm miopen step", lambda: net.rnn.rnn(x.unsqueeze(0), (h[0].unsqueeze(0), h[1].unsqueeze(0))))
    t("addmm 4096x149x1024", lambda: torch.addmm(net.rnn.rnn.bias_ih_l0, x, net.rnn.rnn.weight_ih_l0.t()))
    t("mm 4096x256x1024", lambda: h[0] @ net.rnn.rnn.weight_hh_l0.t())
    f = torch.rand(B, 256, device=dev)
    t("actor+critic heads", lambda: (net.actor(f).logits, net.critic(f)))
    t("dist sample", lambda: net.actor(f).sample())
    t("one_hot embed", lambda: net.scalar_embed(F.one_hot(d.long(), 4).squeeze(1).float()))
    t("full act()", lambda: net.act({"image": img, "direction": d}, h, m))
