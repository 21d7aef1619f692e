import os, sys, torch
sys.path.insert(0, os.getcwd())
from wavenet_amd import modem as M
m = M.Modem(M.esquema_b())
S = 2000
stride = M.pad4(m.esquema.max_frame_samples(126)); nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8); fields[:, 0] = 1; fields[:, 13] = 3; fields[:, 14] = 107
fields[:, 1:13] = torch.randint(0, 256, (S, 12), dtype=torch.uint8, generator=g)
frames, lens = m.build_frames(fields.cuda(), torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g).cuda(), 128)
audio, _ = m.encode_streams(frames, lens, stride)
m.detect_stats()
m.detect(audio, S, stride, nwin); torch.cuda.synchronize()
st = m.detect_stats(paths=True)
w = st[0]
print("windows %d: long-way passes %.4f, of them two-tile %.4f, sixteen chosen columns %.4f, separator bin alone %.4f, another tone's bin alone %.4f, done twice %.4f, s1 units %.1f, s2 cells %.1f" % (w, st[6]/w, st[7]/w, st[8]/w, st[9]/w, st[10]/w, st[5]/w, st[4]/w, st[3]/w))
