import sys, torch
sys.path.insert(0, '.')
from spacer_amd import kernels as K
dev = torch.device('cuda:0'); BF = torch.bfloat16
def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(*shape, generator=g)*scale).to(BF).to(dev)
D, Hq, Hkv, B, nP, Pmax, Cmax = 128, 6, 1, 5, 2, 70, 16
q = rnd((B, Hq*D), 1, .7); pk, pv = rnd((nP,Pmax,Hkv,D),2,.7), rnd((nP,Pmax,Hkv,D),3,.7); tk, tv = rnd((B,Cmax,Hkv,D),4,.7), rnd((B,Cmax,Hkv,D),5,.7)
plen = torch.tensor([70,33], dtype=torch.int32, device=dev); pof = torch.tensor([0,0,1,1,1], dtype=torch.int32, device=dev)
for tl in (0, 9):
    tld = torch.tensor([tl], dtype=torch.int32, device=dev)
    o = K.attn_decode(q, pk, pv, plen, pof, tk, tv, tld, Hq, Hkv, D, D**-0.5)
    for b in range(B):
        P = int(plen[pof[b]])
        kk = torch.cat([pk[pof[b], :P], tk[b, :tl+1]]).float().repeat_interleave(Hq//Hkv, 1)
        vv = torch.cat([pv[pof[b], :P], tv[b, :tl+1]]).float().repeat_interleave(Hq//Hkv, 1)
        s = torch.einsum("hd,lhd->hl", q[b].float().view(Hq, D), kk) * D**-0.5
        want = torch.einsum("hl,lhd->hd", torch.softmax(s, -1), vv)
        err = (o[b].float().view(Hq, D) - want).abs()
        bad = (err > 0.05).nonzero()
        print(tl, b, 'total keys', P+tl+1, 'nbad', len(bad), bad[:12].tolist(), [float(o[b].view(Hq,D)[i,j]) for i,j in bad[:4].tolist()])
