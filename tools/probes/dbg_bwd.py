import sys, torch, ctypes as C
sys.path.insert(0, "/root/repo")
from tpnet_amd import _lib, fused_feature as ff
n = 5000
torch.manual_seed(n)
mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).cuda()
x = torch.rand(n, 64, device="cuda") * 9.0
x[:, 5] = 0.0
gy = torch.randn(n, 64, device="cuda")
for p in mlp.parameters(): p.grad = None
y = ff.mlp_f32(mlp, x); y.backward(gy)
g_mod = [p.grad.clone().double() for p in mlp.parameters()]
for p in mlp.parameters(): p.grad = None
mlp(x).backward(gy)
g_ref = [p.grad.clone().double() for p in mlp.parameters()]
w1, b1, w2 = mlp[0].weight.detach().double(), mlp[0].bias.detach().double(), mlp[2].weight.detach().double()
xd, gd = x.double(), gy.double()
pre = xd @ w1.t() + b1
on = (pre > 0).double(); hid = pre * on; gh = (gd @ w2) * on
want = [gh.t() @ xd, gh.sum(0), gd.t() @ hid, gd.sum(0)]
for name, a, b, w in zip(("w1", "b1", "w2", "b2"), g_mod, g_ref, want):
    print(name, "mod-ref", float((a - b).abs().max()), "mod-want", float((a - w).abs().max()), "ref-want", float((b - w).abs().max()), "scale", float(w.abs().max()))
# direct call on the same inputs
prep = ff.prepared(mlp, 64)
lib = _lib.load()
pf = int(lib.tpnet_mlp64_bwd_partial_floats()); nblk = min(256, (n + 31) // 32)
part = torch.empty((nblk, pf), device="cuda")
rc = lib.tpnet_mlp64_bwd_f32(x.data_ptr(), gy.data_ptr(), n, prep[2], part.data_ptr(), nblk, C.c_void_p(torch.cuda.current_stream().cuda_stream))
tot = part[:rc].sum(0).double()
gw1 = tot[:256 * 64].view(256, 64)
print("direct gw1 - want", float((gw1 - want[0]).abs().max()), "direct - autograd path", float((gw1 - g_mod[0]).abs().max()))
near = (pre.abs() < 1e-3)
print("near-zero pre-activations:", int(near.sum()), "of", pre.numel())
d = (gw1 - want[0]).abs(); i = int(d.argmax()); u = i // 64
print("worst row", u, "its near count", int(near[:, u].sum()), "min |pre| in that unit", float(pre[:, u].abs().min()))
near = (pre.abs() < 1e-4 * (1.0 + xd.abs().max())).double()
gha = (gd @ w2).abs()
mags = [((gha * on).t() @ xd.abs(), (gha * near).t() @ xd.abs()), ((gha * on).sum(0), (gha * near).sum(0)),
        (gd.abs().t() @ hid.abs(), gd.abs().t() @ (pre.abs() * near)), (gd.abs().sum(0), torch.zeros(64, device="cuda", dtype=torch.float64))]
for name, a, b, (m, amb) in zip(("w1", "b1", "w2", "b2"), g_mod, g_ref, mags):
    tol = 5e-5 * m + 2 * amb + 1e-5
    r = (a - b).abs() / tol
    i = int(r.argmax())
    print(name, "max ratio", float(r.max()), "at", i, "diff", float((a - b).abs().flatten()[i]), "m", float(m.flatten()[i]), "amb", float(amb.flatten()[i]))
