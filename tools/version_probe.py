"""Do optimiser steps bump Parameter._version?  (The inference handle of the mirror keys on (data_ptr, _version).)"""
import torch
dev = "cuda:0"
for fused in (False, True):
    for foreach in ((None,) if fused else (False, True)):
        for scaler_on in (False, True):
            p = torch.nn.Parameter(torch.randn(1000, device=dev))
            kw = dict(fused=True) if fused else dict(foreach=foreach)
            opt = torch.optim.AdamW([p], lr=1e-3, **kw)
            scaler = torch.amp.GradScaler("cuda") if scaler_on else None
            v0, d0 = p._version, p.data_ptr()
            before = p.detach().clone()
            loss = (p * p).sum()
            if scaler is not None:
                scaler.scale(loss).backward(); scaler.step(opt); scaler.update()
            else:
                loss.backward(); opt.step()
            torch.cuda.synchronize()
            print(f"fused={fused} foreach={foreach} scaler={scaler_on}: version {v0} -> {p._version}, data_ptr same {p.data_ptr() == d0}, values changed {not torch.equal(before, p.detach())}")
