import sys, os, warnings, numpy as np, torch
sys.path.insert(0, os.getcwd())
import lhotse_amd as LA
for sr in (22050, 24000, 48000, 24000):
    S = 10 * sr
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = LA.HipFbank(LA.HipFbankConfig(sampling_rate=sr))
    plan = ex.plan
    wave = torch.empty(4000 * S, device="cuda").uniform_(-0.5, 0.5)
    offs = np.arange(4000, dtype=np.int64) * S
    lens = np.full(4000, S, dtype=np.int64)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(14)]
    for s, e in evs:
        s.record(); out, fr = plan.run(wave, offs, lens, None); e.record()
    torch.cuda.synchronize()
    print(sr, plan.kernel_name.split(" lds")[0], [round(s.elapsed_time(e), 2) for s, e in evs], flush=True)
    del wave, out
