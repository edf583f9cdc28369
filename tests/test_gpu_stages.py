"""GPU: stage graphs of the engine (encode_key / encode_value / decode / fusion) against the oracle on the
same seeded inputs, and against the reference-captured goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, sample_of
from eva_vos_amd import synth
from gpu_util import call, dev, model_handle, rows_to_nchw, stream
from oracle import stcn_oracle as O

pytestmark = pytest.mark.gpu
STAGE = {"stA": (128, 160, 1), "stB": (100, 150, 3), "stC": (96, 208, 2)}


def gpu_encode_key(nets, img):
    nh, nw = img.shape[-2:]
    h, w = nh // 16, nw // 16
    o = dict(k16=torch.empty(1, h * w, 64, device="cuda"), f16_thin=torch.empty(1, h * w, 512, device="cuda"),
             f16=torch.empty(1, h * w, 1024, device="cuda"), f8=torch.empty(1, 4 * h * w, 512, device="cuda"),
             f4=torch.empty(1, 16 * h * w, 256, device="cuda"))
    call("stcn_test_encode_key", model_handle(nets), stream(), dev(img), nh, nw, o["k16"], o["f16_thin"],
         o["f16"], o["f8"], o["f4"])
    return o


def as_nchw(o, nh, nw):
    h, w = nh // 16, nw // 16
    return [rows_to_nchw(o["k16"], h, w), rows_to_nchw(o["f16_thin"], h, w), rows_to_nchw(o["f16"], h, w),
            rows_to_nchw(o["f8"], 2 * h, 2 * w), rows_to_nchw(o["f4"], 4 * h, 4 * w)]


@pytest.mark.parametrize("tag", list(STAGE))
def test_stage_graphs(tag, nets, weights):
    H, W, k = STAGE[tag]
    g = load_golden(tag)
    fw = O.fold_bn(weights[0])
    imgs, _ = O.pad16(synth.synthetic_clip(3, H, W))
    msk = synth.synthetic_mask(3, H, W, k)
    m0, _ = O.pad16(msk[:, 0])
    nh, nw = imgs.shape[-2:]
    h, w = nh // 16, nw // 16
    # ---- encode_key: vs oracle (full tensors) and vs the reference golden samples
    okf = O.encode_key(fw, imgs[:, 0])
    gkf_dev = gpu_encode_key(nets, imgs[:, 0])
    gkf = as_nchw(gkf_dev, nh, nw)
    for n, a, b in zip(["k16", "f16_thin", "f16", "f8", "f4"], gkf, okf):
        assert rel_err(a.numpy(), b.numpy()) < 2e-5, n
        stride = 1 if n == "k16" else 37
        assert rel_err(sample_of(a, stride), g[f"{tag}.key0.{n}.sample"]) < 2e-5, n
    # ---- encode_value
    ov = O.encode_value(fw, imgs[:, 0], okf[2], m0)
    gv = torch.empty(k, h * w, 512, device="cuda")
    call("stcn_test_encode_value", model_handle(nets), stream(), dev(imgs[:, 0]), gkf_dev["f16"],
         dev(m0.reshape(k, -1)), k, nh, nw, gv)
    gv_n = rows_to_nchw(gv, h, w)
    assert rel_err(gv_n.numpy(), ov.numpy()) < 3e-5
    assert rel_err(sample_of(gv_n, 11), g[f"{tag}.value0.sample"]) < 3e-5
    # ---- decode on an oracle readout (isolates the decoder graph)
    okf2 = O.encode_key(fw, imgs[:, 2])
    mk = okf[0].flatten(2).transpose(1, 2)[0]
    mv = ov.flatten(2).transpose(1, 2).contiguous()
    _, _, ro = O.memory_read(mk, mv, okf2[0].flatten(2).transpose(1, 2)[0])          # [k,Q,512]
    oprob, ol4 = O.decode(fw, ro.transpose(1, 2).reshape(k, 512, h, w), okf2[1], okf2[3], okf2[4])
    oagg = O.aggregate(oprob)
    rows = lambda x: dev(x.flatten(2).transpose(1, 2))   # noqa: E731
    l4 = torch.empty(k, 16 * h * w, device="cuda")
    agg = torch.empty(k + 1, nh * nw, device="cuda")
    call("stcn_test_decode", model_handle(nets), stream(), dev(ro), rows(okf2[1]), rows(okf2[3]),
         rows(okf2[4]), k, nh, nw, l4, agg)
    assert (l4.cpu().reshape(ol4.shape) - ol4).abs().max() < 2e-4 * max(1.0, ol4.abs().max().item())
    d = (agg.cpu().reshape(oagg.shape) - oagg).abs().numpy()
    assert np.quantile(d, 0.999) < 1e-3          # saturated multi-object pixels are ill-conditioned
    if k == 1:
        assert d.max() < 1e-3


@pytest.mark.parametrize("tag", list(STAGE))
def test_fusion_net(tag, nets, weights):
    H, W, _ = STAGE[tag]
    g = load_golden(tag)
    imgs, _ = O.pad16(synth.synthetic_clip(2, H, W))
    rng = np.random.Generator(np.random.Philox(key=[7, 7]))
    nh, nw = imgs.shape[-2:]
    prev = torch.from_numpy(rng.uniform(0, 1, (1, 1, nh, nw)).astype(np.float32))
    curr = torch.from_numpy(rng.uniform(0, 1, (1, 1, nh, nw)).astype(np.float32))
    attn = torch.from_numpy(rng.uniform(0, 0.2, (1, 2, nh, nw)).astype(np.float32))
    ref = O.fusion_net(O.fold_bn(weights[1]), imgs[:, 1], prev, curr, attn, 0.25, 0.75)
    out = torch.empty(nh * nw, device="cuda")
    call("stcn_test_fusion", model_handle(nets), stream(), dev(imgs[:, 1]), dev(prev), dev(curr),
         dev(attn), 0.25, 0.75, nh, nw, out)
    assert (out.cpu().reshape(ref.shape) - ref).abs().max() < 2e-4
    assert np.abs(sample_of(out, 13) - g[f"{tag}.fusion_logit.sample"]).max() < 2e-4


def test_full_res_key_encoder_checksums(nets):
    """480x854 (padded 480x864): engine key features against checksums captured from the reference."""
    g = load_golden("st480")
    imgs, _ = O.pad16(synth.synthetic_clip(3, 480, 854))
    o = gpu_encode_key(nets, imgs[:, 0])
    for n in ["k16", "f16_thin", "f16", "f8", "f4"]:
        a = o[n].cpu().numpy().astype(np.float64).reshape(-1)
        mom = np.array([a.sum(), np.abs(a).sum(), (a ** 2).sum()])
        ref = g[f"st480.key0.{n}.moments"]
        assert np.abs(mom[1:] - ref[1:]).max() / ref[1:].max() < 2e-5, n
