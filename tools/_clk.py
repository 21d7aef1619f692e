import os, sys, torch
sys.path.insert(0, os.getcwd())
from wavenet_amd import modem as M
m = M.Modem(M.esquema_b())
S=2000; stride=M.pad4(m.esquema.max_frame_samples(126)); nwin=stride//7055
audio=(0.3*torch.randn((S,stride),device='cuda'))
for i in range(3):
    cls,mag,kbin=m.detect(audio,S,stride,nwin)
torch.cuda.synchronize()
print("memtime/256, memrealtime/256:", kbin.view(-1)[:2].tolist(), "clock GHz =", kbin.view(-1)[0].item()/kbin.view(-1)[1].item()*0.1)
