import os, sys, torch
sys.path.insert(0, os.getcwd())
from wavenet_amd import modem as M
m = M.Modem(M.esquema_b())
S = 1000
stride = M.pad4(m.esquema.max_frame_samples(126)); nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8); fields[:, 0] = 1; fields[:, 13] = 3; fields[:, 14] = 107
fields[:, 1:13] = torch.randint(0, 256, (S, 12), dtype=torch.uint8, generator=g)
frames, lens = m.build_frames(fields.cuda(), torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g).cuda(), 128)
audio, _ = m.encode_streams(frames, lens, stride)
cls, mag, kbin = m.detect(audio, S, stride, nwin)
x = audio[:, :nwin * 7055].reshape(S, nwin, 7055).double()
e_half = 0.5 * (7055 * (x * x).sum(-1) + x.sum(-1) ** 2)
share = mag.double() ** 2 / e_half.clamp_min(1e-30)
tot = S * nwin
for b, name in ((32, "separator"), (40, "ping"), (48, "end")):
    sel = kbin == b
    print(name, "windows with that bin: %.4f" % (sel.sum().item() / tot),
          " share of the half-spectrum energy in it >= 0.5/0.55/0.6/0.7/0.9: " +
          " ".join("%.4f" % ((sel & (share >= t)).sum().item() / tot) for t in (0.5, 0.55, 0.6, 0.7, 0.9)))
sel = (kbin != 32) & (kbin != 40) & (kbin != 48)
print("other bins %.4f; of all windows with share >= 0.5/0.7/0.9: " % (sel.sum().item() / tot) + " ".join("%.4f" % ((sel & (share >= t)).sum().item() / tot) for t in (0.5, 0.7, 0.9)))
print("silent windows (energy 0): %.4f" % ((e_half == 0).sum().item() / tot))
