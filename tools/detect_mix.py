import os, sys, torch
sys.path.insert(0, os.getcwd())
from wavenet_amd import modem as M
m = M.Modem(M.esquema_b())
S = 2000
stride = M.pad4(m.esquema.max_frame_samples(126)); nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8); fields[:, 0] = 1
fields[:, 1:13] = torch.randint(0, 256, (S, 12), dtype=torch.uint8, generator=g); fields[:, 13] = 3; fields[:, 14] = 107
frames, lens = m.build_frames(fields.cuda(), torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g).cuda(), 128)
clean, _ = m.encode_streams(frames, lens, stride)
for what, audio in (("clean", clean), ("noisy 0.1", clean + 0.1 * torch.randn_like(clean)), ("noise", 0.3 * torch.randn_like(clean))):
    m.detect_stats()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    m.detect(audio, S, stride, nwin); torch.cuda.synchronize(); m.detect_stats()
    e0.record(); m.detect(audio, S, stride, nwin); e1.record(); torch.cuda.synchronize()
    st = m.detect_stats(columns=True)
    print(what, "ms %.3f" % e0.elapsed_time(e1), "stage-2 units/win %.2f stage-1 units/win %.1f redone %.4f" % (st[3]/st[0], st[4]/st[0], st[5]/st[0]), st[:3])
