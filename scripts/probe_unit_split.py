"""Developer probe: the row-split MaCowUnit kernels (csrc/mcf_unit_split.hip) against the one-workgroup-per-sample kernels --
bit-identity of every output on random operands, repeated with a copy stream running beside them, and isolated timing with
rotating (L2-cold) weight sets.  usage: probe_unit_split.py [C] [B] [fwd|both]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import _lib, ops
from ipoke_amd._lib import check

CT = int(sys.argv[1]) if len(sys.argv) > 1 else 64
BT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
WHAT = sys.argv[3] if len(sys.argv) > 3 else "both"
dt, tdt, dev = "bf16", torch.bfloat16, "cuda"
lib = _lib.lib(); s = _lib.current_stream()
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape, scale=0.05):
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(tdt)


class Case:
    def __init__(self, C, B, ld, NU):
        self.C, self.B, self.ld, self.NU = C, B, ld, NU
        dm = self.dm = ops.mcf_dims(C, 128, dt)
        M = self.M = B * 64
        self.x = torch.randn(M, ld, device=dev, generator=g)
        self.cond = torch.randn(M, 128, device=dev, generator=g).to(tdt)
        self.dy = torch.randn(M, ld, device=dev, generator=g)
        self.dld = torch.randn(B, device=dev, generator=g)
        self.W = [[dict(W1=rnd(dm["Hr"], dm["K1p"]), W2=rnd(dm["N2r"], dm["K2p"]), W1T=rnd(dm["Cr"], 6 * dm["Hq"]), W2T=rnd(dm["Hr"], dm["K3p"]))
                   for _ in range(4)] for _ in range(NU)]
        self.bias2 = torch.randn(2 * C, device=dev, generator=g) * 0.1
        self.pls = torch.randn(C, device=dev, generator=g) * 0.1
        self.pb = torch.randn(C, device=dev, generator=g) * 0.1

    def outputs(self, S):
        C, B, M, ld, dm = self.C, self.B, self.M, self.ld, self.dm
        o = dict(ys=[torch.full((M, ld), float("nan"), device=dev) for _ in range(4)],
                 a2=[torch.zeros(M, dm["K2p"], device=dev, dtype=tdt) for _ in range(4)],
                 sc=[torch.zeros(M, C, device=dev) for _ in range(4)],
                 dps=[torch.zeros(M, dm["K3p"], device=dev, dtype=tdt) for _ in range(4)],
                 dcs=[torch.zeros(M, dm["Hq"], device=dev, dtype=tdt) for _ in range(4)],
                 xop=[torch.zeros(M, dm["Cp"], device=dev, dtype=tdt) for _ in range(4)],
                 dbp=[torch.zeros(B * S, 2 * C, device=dev) for _ in range(4)],
                 pp=[torch.zeros(B * S, 2 * C, device=dev) for _ in range(4)],
                 slot=torch.zeros(4, B, 4, device=dev), dx=torch.full((M, ld), float("nan"), device=dev),
                 zc=torch.full((M, 40), 3.0, device=dev, dtype=tdt))
        nb = lib.ipoke_macow_unit_xchg_bytes(B, S)
        o["xchg"] = torch.zeros(max(nb, 8) // 4, dtype=torch.int32, device=dev)
        return o

    def descs(self, u, o, S, saved=None):
        """saved: outputs of a forward run whose saves the backward call reads (default: o itself)"""
        sv = saved or o
        C, B, ld = self.C, self.B, self.ld
        d4 = (_lib.McfDesc * 4)()
        ins = [self.x, sv["ys"][0], sv["ys"][1], sv["ys"][2]]
        for k in range(4):
            d = d4[k]
            d.ld, d.C, d.B, d.cond, d.Cc, d.order, d.rows_per_block = ld, C, B, self.cond.data_ptr(), 128, k, 16
            w = self.W[u][k]
            d.W1, d.W2, d.W1T, d.W2T, d.bias2 = w["W1"].data_ptr(), w["W2"].data_ptr(), w["W1T"].data_ptr(), w["W2T"].data_ptr(), self.bias2.data_ptr()
            d.x = ins[k].data_ptr(); d.y = o["ys"][k].data_ptr(); d.a2_save = sv["a2"][k].data_ptr(); d.scale_save = sv["sc"][k].data_ptr()
            d.logdet_slot = o["slot"][k].data_ptr()
            d.dparams_save = o["dps"][k].data_ptr(); d.dc_save = o["dcs"][k].data_ptr(); d.dbias_part = o["dbp"][k].data_ptr()
            d.x_op_save = o["xop"][k].data_ptr()
            if k in (1, 3):
                d.post_log_scale, d.post_bias, d.y_post, d.post_part = self.pls.data_ptr(), self.pb.data_ptr(), sv["ys"][k].data_ptr(), o["pp"][k].data_ptr()
        d4[3].dy = self.dy.data_ptr(); d4[0].dx = o["dx"].data_ptr(); d4[0].dld = self.dld.data_ptr()
        d4[3].zc_out, d4[3].zc_off, d4[3].zc_stride, d4[3].zc_cin, d4[3].zc_ld = o["zc"].data_ptr(), 1, 2, C // 2, 40
        if S > 1:
            d4[0].split = S; d4[0].xchg = o["xchg"].data_ptr()
        return d4


def check_case(C, B, ld, reps=6, bwd=True):
    cs = Case(C, B, ld, 1)
    ref = cs.outputs(1)
    check(lib.ipoke_macow_unit_fwd(cs.descs(0, ref, 1), _lib.BF16, s))
    if bwd:
        check(lib.ipoke_macow_unit_bwd(cs.descs(0, ref, 1), _lib.BF16, s))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
    for S in (2, 4):
        for rep in range(reps):
            o = cs.outputs(S)
            if rep % 2:                        # uneven load beside the hand-offs
                with torch.cuda.stream(side):
                    for _ in range(3):
                        big2.copy_(big)
            check(lib.ipoke_macow_unit_fwd(cs.descs(0, o, S), _lib.BF16, s))
            if bwd:
                check(lib.ipoke_macow_unit_bwd(cs.descs(0, o, S, saved=ref), _lib.BF16, s))
            torch.cuda.synchronize()
            bad = []
            names = ["ys", "a2", "sc"] + (["dps", "dcs", "xop"] if bwd else [])
            for nme in names:
                for k in range(4):
                    cols = C if nme == "ys" and k < 3 else None        # pass-through columns of the intermediate states are not written
                    if not torch.equal(o[nme][k][:, :cols], ref[nme][k][:, :cols]):
                        bad.append((nme, k, (o[nme][k].float() - ref[nme][k].float()).abs().max().item()))
            if not torch.equal(o["zc"], ref["zc"]):
                bad.append(("zc",))
            if bwd and not torch.equal(o["dx"], ref["dx"]):
                bad.append(("dx", (o["dx"] - ref["dx"]).abs().max().item()))
            e_ld = (o["slot"].sum(2) - ref["slot"].sum(2)).abs().max().item()
            e_db = e_pp = 0.0
            if bwd:
                for k in range(4):
                    rs = ref["dbp"][k].sum(0); e_db = max(e_db, ((o["dbp"][k].sum(0) - rs).abs().max() / rs.abs().max().clamp_min(1e-6)).item())
                    rp = ref["pp"][k].sum(0); e_pp = max(e_pp, ((o["pp"][k].sum(0) - rp).abs().max() / rp.abs().max().clamp_min(1e-6)).item())
            tmo = int(o["xchg"][0].item()); dirty = int((o["xchg"][64:] != 0).sum().item())
            ok = not bad and e_ld < 1e-3 and e_db < 1e-5 and e_pp < 1e-5 and tmo == 0 and dirty == 0
            if not ok or rep == 0:
                print(f"C={C} ld={ld} B={B} S={S} rep={rep}: {'ok' if ok else 'MISMATCH'} bad={bad[:6]} logdet {e_ld:.2e} dbias {e_db:.1e} "
                      f"post {e_pp:.1e} timeouts {tmo} dirty granules {dirty}")
            if not ok:
                return False
    return True


def timing(C, B, n=240, bwd=True):
    NU = 24
    cs = Case(C, B, 64, NU)
    for S in (1, 2, 4):
        o = cs.outputs(S)
        D = [cs.descs(u, o, S) for u in range(NU)]
        res = []
        for fn in [lib.ipoke_macow_unit_fwd] + ([lib.ipoke_macow_unit_bwd] if bwd else []):
            for u in range(NU):
                check(fn(D[u], _lib.BF16, s))
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                check(fn(D[i % NU], _lib.BF16, s))
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / n * 1e3)
        print(f"timing C={C} B={B} split={S}: " + "  ".join(f"{t:.1f} us" for t in res) + f"  (fwd{', bwd' if bwd else ''}; timeouts {int(o['xchg'][0].item())})")


def stamps(C, B, S):
    import ctypes
    probe_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libunit_probe.so")
    if not os.path.exists(probe_path):
        return
    P = ctypes.CDLL(probe_path)
    P.ipoke_macow_unit_fwd.argtypes = [ctypes.POINTER(_lib.McfDesc), ctypes.c_int, ctypes.c_void_p]
    P.ipoke_macow_unit_set_stamps.argtypes = [ctypes.c_void_p]
    NU = 24
    cs = Case(C, B, 64, NU)
    o = cs.outputs(S)
    st = torch.zeros(256, dtype=torch.int64, device=dev)
    P.ipoke_macow_unit_set_stamps(ctypes.c_void_p(st.data_ptr()))
    for u in range(NU):
        assert P.ipoke_macow_unit_fwd(cs.descs(u, o, S), _lib.BF16, s) == 0
    torch.cuda.synchronize()
    t = st.cpu().view(4, 64).tolist()
    names = ["entry", "staged"]
    for k in range(4):
        names += [f"L{k} tile(s) a", f"L{k} halo+sync", f"L{k} last tile", f"L{k} sync", f"L{k} a2save", f"L{k} gemm2", f"L{k} sync", f"L{k} coupling+sum"]
    t00 = t[0][0]
    print(f"forward stamps C={C} B={B} S={S} (cycles @100 MHz s_memtime? see ratio; columns = workgroups 0..{S - 1}: delta / absolute)")
    for i, nme in enumerate(names):
        row = "  ".join(f"+{t[w][i] - t[w][i - 1] if i else 0:6d} ({t[w][i] - t00:7d})" for w in range(S))
        print(f"  {nme:18s} {row}")


if __name__ == "__main__":
    bwd = WHAT != "fwd"
    allok = True
    for (C, ld, B) in [(64, 64, 5), (60, 64, 3), (32, 64, 4), (30, 32, 3), (8, 8, 2), (64, 64, 20)]:
        allok &= check_case(C, B, ld, bwd=bwd)
    print("ALL OK" if allok else "FAILURES")
    timing(CT, BT, bwd=bwd)
    timing(32, BT, bwd=bwd)
    for S in (2, 4):
        stamps(CT, BT, S)
