#!/usr/bin/env python3
"""Round-4 diagnostic (GPU box): on the QUANTISED config-2 batch (boundary plane rounded to k/100), how much of a frame sits in
union-find components the parallel flood cannot resolve, and could those components be flooded one by one?

Round-3 review, item 2: "count unresolved union-find components, and how many of them contain two SEEDS OF EQUAL VALUE -- only
those are coupled to the global heap" (a component of the minimum-level links is closed under pushes; with a local (value,
age) heap its pop order is reproduced unless two of its seeds tie in value, because the order of equal-valued seeds is
defined by the layout of the reference's one binary heap).

Method: the watershed runs in mode 2 (parallel levels only) on the first frames of the batch; the minimax levels L are read
out of the call's workspace (layout of pcseg_watershed4_f32's Carver) and the first-level components are rebuilt on the host
exactly as ws_uf_tile_frame defines them: every reachable non-seed pixel is linked to ALL its 4-neighbours whose level equals
the minimum neighbour level; a component holding two different marker ids is unresolved.  No oracle involved."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from particle_col_image_segmentation_amd import _lib, ops, synth

INF = 0xFFFFFFFF


def al(x, a=256):
    return (x + a - 1) // a * a


def components(L, markers, mask):
    H, W = L.shape
    n = H * W
    Lp = np.full((H + 2, W + 2), INF, np.uint64)
    Lp[1:-1, 1:-1] = L
    nb = np.stack([Lp[:-2, 1:-1], Lp[1:-1, :-2], Lp[1:-1, 2:], Lp[2:, 1:-1]])  # up, left, right, down
    m = nb.min(0)
    seed = (markers != 0) & (mask != 0)
    reach = L != INF
    src = reach & ~seed & (m != INF)
    idx = np.arange(n).reshape(H, W)
    offs = [-W, -1, 1, W]
    rows, cols = [], []
    for k in range(4):
        sel = src & (nb[k] == m)
        rows.append(idx[sel])
        cols.append(idx[sel] + offs[k])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    g = coo_matrix((np.ones(rows.size, np.uint8), (rows, cols)), shape=(n, n))
    ncomp, lab = connected_components(g, directed=False)
    return lab.reshape(H, W), reach, seed


def main():
    dev = torch.device("cuda", 0)
    B, H, W = 8, 1024, 1024
    n_frames_host = 4
    stack = synth.gen_batch_torch(10_000, B, H, W, dev)
    lib = _lib.load()
    for levels in (0, 100):
        if levels:
            stack[:, 3] = torch.round(stack[:, 3] * levels) / levels
        bm = stack[:, 3]
        d2, mask = ops.edt_sq_lt(bm, 0.5)
        _, markers, n_markers = ops.local_maxima(d2, want_mask=False)
        img, fstride = ops._plane_view(bm)
        out = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        flags_out = torch.zeros((B,), dtype=torch.int32, device=dev)
        nbytes = lib.pcseg_watershed_workspace_bytes(B, H, W)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.pcseg_watershed4_f32(ops._ptr(img), fstride, ops._ptr(markers), ops._ptr(mask), ops._ptr(out), ops._ptr(flags_out),
                                            B, H, W, 2, ops._ptr(ws), nbytes, ops._stream()), "watershed")
        torch.cuda.synchronize()
        n = B * H * W
        o_val, o_L = 0, al(n * 4)
        val_all = ws[o_val:o_val + 4 * n].view(torch.int32).cpu().numpy().view(np.uint32).reshape(B, H, W)
        L_all = ws[o_L:o_L + 4 * n].view(torch.int32).cpu().numpy().view(np.uint32).reshape(B, H, W)
        mk, ms, lab_out = markers.cpu().numpy(), mask.cpu().numpy(), out.cpu().numpy()
        print("=== levels=%d: frames flagged after both parallel levels: %d / %d" % (levels, int(flags_out.sum()), B), flush=True)
        for b in range(n_frames_host):
            comp, reach, seed = components(L_all[b], mk[b], ms[b])
            cid = comp[reach]
            # marker ids per component
            sc, sm, sv = comp[seed], mk[b][seed], val_all[b][seed]
            order = np.lexsort((sm, sc))
            sc, sm, sv = sc[order], sm[order], sv[order]
            first = np.r_[True, sc[1:] != sc[:-1]]
            newid = np.r_[True, (sc[1:] != sc[:-1]) | (sm[1:] != sm[:-1])]
            comp_of_group = sc[newid]                    # one entry per (component, marker id)
            ids_per_comp = np.bincount(comp_of_group, minlength=comp.max() + 1)
            bad = ids_per_comp >= 2                      # unresolved at the first level
            sizes = np.bincount(cid, minlength=comp.max() + 1)
            # two equal-valued seed PIXELS inside one component -- also of one marker id: which of them pops first decides the
            # ages of their neighbours, and through them ties against entries of other labels further out
            o2 = np.lexsort((sv, sc))
            c2, v2 = sc[o2], sv[o2]
            dup = np.zeros(comp.max() + 1, bool)
            same = (c2[1:] == c2[:-1]) & (v2[1:] == v2[:-1])
            dup[c2[1:][same]] = True
            nbad = int(bad.sum())
            px_reach = int(reach.sum())
            px_bad = int(sizes[bad].sum())
            free = bad & ~dup
            unl = int(((lab_out[b] == 0) & reach).sum())
            print("frame %d: reachable px %d, seeds (marker ids) %d, first-level components with a seed %d" %
                  (b, px_reach, int(newid.sum()), int((ids_per_comp >= 1).sum())))
            print("   unresolved components %d holding %d px = %.1f %% of the reachable pixels; marker ids inside them %d" %
                  (nbad, px_bad, 100.0 * px_bad / max(px_reach, 1), int(ids_per_comp[bad].sum())))
            if nbad:
                s = np.sort(sizes[bad])[::-1]
                print("   sizes (px): max %d, top-5 %s, median %d; marker ids per unresolved component: max %d median %d" %
                      (s[0], s[:5].tolist(), int(np.median(s)), int(ids_per_comp[bad].max()), int(np.median(ids_per_comp[bad]))))
                print("   WITH two equal-valued seeds (coupled to the global heap): %d components, %d px (%.1f %% of the unresolved px)" %
                      (int((bad & dup).sum()), int(sizes[bad & dup].sum()), 100.0 * sizes[bad & dup].sum() / max(px_bad, 1)))
                print("   WITHOUT (locally floodable): %d components, %d px (%.1f %%), largest %d px" %
                      (int(free.sum()), int(sizes[free].sum()), 100.0 * sizes[free].sum() / max(px_bad, 1),
                       int(sizes[free].max()) if free.any() else 0))
            print("   px still unlabelled after the second parallel level: %d (%.1f %% of reachable)" % (unl, 100.0 * unl / max(px_reach, 1)), flush=True)


if __name__ == "__main__":
    main()
