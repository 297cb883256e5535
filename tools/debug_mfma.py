import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from meta_learning_pacoh_amd import _lib as L
    torch.manual_seed(0)
    for (T, P, n, f) in [(4, 3, 5, 2), (2, 2, 16, 2), (2, 2, 17, 2), (2, 2, 32, 2), (2, 2, 48, 2), (3, 4, 64, 2), (3, 4, 64, 4)]:
        g = torch.Generator().manual_seed(n + f)
        B = T * P
        z = torch.randn(B, n, f, generator=g); mean = 0.3 * torch.randn(B, n, generator=g); y = torch.randn(T, n, generator=g)
        ls = torch.nn.functional.softplus(torch.randn(P, f, generator=g)); os_ = torch.nn.functional.softplus(torch.randn(P, generator=g))
        noise = torch.nn.functional.softplus(torch.randn(P, generator=g) - 1)
        out = L.gp_lml_fwdbwd(z.cuda(), 1, mean.cuda(), L.MEAN_VECTOR, y.cuda(), P, ls.cuda(), os_.cuda(), noise.cuda(), B, P)
        torch.cuda.synchronize()
        print(n, f, 'lml', out[0].cpu().numpy().round(5).tolist()[:6], 'info', out[6].cpu().tolist(), 'dls', out[3].cpu().numpy().round(4).tolist()[:2], 'dnoise', out[5].cpu().numpy().round(5).tolist()[:4], 'dz', out[1].cpu()[0, :2].numpy().round(5).tolist(), 'dm', out[2].cpu()[0, :3].numpy().round(5).tolist())
else:
    for dis in ('1', '0'):
        env = dict(os.environ, PACOH_DISABLE_MFMA=dis)
        print('--- PACOH_DISABLE_MFMA=' + dis, flush=True)
        subprocess.run([sys.executable, __file__, 'child'], env=env)
