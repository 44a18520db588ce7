import sys, os
sys.path.insert(0, os.getcwd())
import torch, rnnt_amd
dev = "cuda:0"
def tryit(name, fn):
    try:
        r = fn()
        print(name, "->", "ok", getattr(r, "shape", r))
    except Exception as e:
        print(name, "->", type(e).__name__, str(e)[:110])
B, T, U, V = 2, 5, 3, 7
acts = torch.randn(B, T, U + 1, V, device=dev)
lab = torch.randint(0, V - 1, (B, U), dtype=torch.int32)
il = torch.tensor([5, 4]); ll = torch.tensor([3, 2])
tryit("bf16 acts", lambda: rnnt_amd.rnnt_loss(acts.bfloat16(), lab, il, ll, 6))
tryit("f64 acts", lambda: rnnt_amd.rnnt_loss(acts.double(), lab, il, ll, 6))
tryit("cpu acts", lambda: rnnt_amd.rnnt_loss(acts.cpu(), lab, il, ll, 6))
tryit("B=0", lambda: rnnt_amd.rnnt_loss(acts[:0], lab[:0], il[:0], ll[:0], 6))
tryit("3-d acts", lambda: rnnt_amd.rnnt_loss(acts[0], lab, il, ll, 6))
tryit("int64 labels, python lists", lambda: rnnt_amd.rnnt_loss(acts, lab.long().tolist(), [5, 4], [3, 2], 6))
tryit("labels wrong U", lambda: rnnt_amd.rnnt_loss(acts, lab[:, :2], il, ll, 6))
tryit("blank = V", lambda: rnnt_amd.rnnt_loss(acts, lab, il, ll, 7))
tryit("U=0", lambda: rnnt_amd.rnnt_loss(acts[:, :, :1].contiguous(), torch.zeros(B, 0, dtype=torch.int32), il, [0, 0], 6))
tryit("transposed view", lambda: rnnt_amd.rnnt_loss(acts.transpose(1, 2).contiguous().transpose(1, 2), lab, il, ll, 6))
tryit("no grad needed", lambda: rnnt_amd.rnnt_loss(acts, lab, il, ll, 6).sum().item())
a2 = acts.clone().requires_grad_(True)
tryit("double backward call", lambda: (lambda c: (c.sum().backward(retain_graph=True), c.sum().backward()))(rnnt_amd.rnnt_loss(a2, lab, il, ll, 6)))
tryit("joint V>4096 via lambda", lambda: rnnt_amd.rnnt_lambda_func([torch.randn(1, 5, 5000, device=dev), torch.randn(1, 2, 5000, device=dev), torch.zeros(1, 1, dtype=torch.int32), torch.tensor([[5]]), torch.tensor([[1]])]).shape)
tryit("joint_dense K=200", lambda: rnnt_amd.joint_dense(torch.randn(4, 200, device=dev), torch.randn(200, 8, device=dev), torch.zeros(8, device=dev)).shape)
tryit("RNNTLoss module", lambda: rnnt_amd.RNNTLoss(blank_label=6)(acts, lab, il, ll))
tryit("second device string", lambda: rnnt_amd.rnnt_loss(acts.to("cuda"), lab, il, ll, 6).shape)
