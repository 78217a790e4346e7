"""A recording fake of libsampt_hip.so for CPU tests (TEST INFRASTRUCTURE ONLY).

``install(monkeypatch, sam_sd, cfg, pips_sd)`` replaces the ctypes layer of ``sam_pt_amd._lib`` — library handle, pointer
helpers and the HIP-device check — by a Python object with the same entry points (include/sampt_hip.h) that receives the
TENSORS themselves instead of device pointers and fills the outputs with the CPU oracle.  The product classes
(``SamPredictor``, ``PipsPointTracker``, ``SamPt``) then run their real host logic on CPU tensors: argument marshalling,
buffer shapes, window chaining, prompt assembly.  Every call is counted in ``fake.calls``.  Nothing in ``sam_pt_amd``
knows about this file; without it a CPU device still raises (tests/test_cpu_host.py::test_product_fails_loudly_without_gpu).
"""
from __future__ import annotations

import contextlib
from collections import Counter

import torch


def _set(ref, value):
    ref._obj.value = value          # ctypes.byref(x) keeps x as _obj


class FakeHip:
    def __init__(self, sam_sd=None, cfg=None, pips_sd=None, hq=False):
        self.sam_sd, self.cfg, self.pips_sd, self.hq = sam_sd, cfg, pips_sd, hq
        self.calls = Counter()
        self._next = 1

    def __getattr__(self, name):      # anything not modelled: fail loudly with the symbol name
        raise AttributeError(f"FakeHip: {name} is not modelled")

    def _handle(self, out):
        _set(out, self._next)
        self._next += 1
        return 0

    def sampt_last_error(self):
        return b"fake"

    # ------------------------------------------------------------------ PIPS
    def sampt_pips_create(self, names, ptrs, n, stride, S, out):
        self.calls["pips_create"] += 1
        assert stride == 4 and S == 8 and "fnet.conv1.weight" in names and "delta_block.to_delta.0.weight" in names
        return self._handle(out)

    def sampt_pips_destroy(self, h):
        self.calls["pips_destroy"] += 1

    def sampt_pips_fnet_workspace_bytes(self, h, nf, H, W, out):
        _set(out, 256)
        return 0

    def sampt_pips_fnet_f32(self, h, frames, nf, H, W, outs, ws, nbytes, stream):
        from oracle import pips_ref as PO
        self.calls["pips_fnet_frames"] += nf
        assert frames.dtype == torch.uint8 and tuple(frames.shape) == (nf, 3, H, W)
        fm = torch.cat([PO.fnet(self.pips_sd, PO.normalize_rgbs(frames[i:i + 1]), 4) for i in range(nf)])
        for lvl, p in enumerate(PO.build_pyramid(fm)):
            assert tuple(outs[lvl].shape) == (nf, p.shape[2], p.shape[3], 128)
            outs[lvl].copy_(p.permute(0, 2, 3, 1))
        return 0

    def sampt_pips_sample_feat_f32(self, fmap, H0, W0, frame_idx, xy, n, out, stream):
        from oracle import pips_ref as PO
        self.calls["pips_sample_feat"] += 1
        for i in range(n):
            f = int(frame_idx[i]) if frame_idx is not None else 0
            out[i] = PO.bilinear_sample2d(fmap[f].permute(2, 0, 1), xy[i:i + 1, 0], xy[i:i + 1, 1])[0]
        return 0

    def sampt_pips_update_workspace_bytes(self, h, n, out):
        _set(out, 256)
        return 0

    def sampt_pips_update_f32(self, h, pyr, H0, W0, fidx, n, xys, feat_init, iters, tr_o, vi_o, ws, nbytes, stream):
        from oracle import pips_ref as PO
        self.calls["pips_update"] += 1
        self.calls["pips_update_points"] += n
        assert tuple(fidx.shape) == (n, 8) and tuple(tr_o.shape) == (8, n, 2) and iters == 6
        groups = {}
        for i in range(n):
            groups.setdefault(tuple(fidx[i].tolist()), []).append(i)
        for frames, idx in groups.items():           # points sharing a window go through one Pips.forward, like the reference
            fm = pyr[0][list(frames)].permute(0, 3, 1, 2)
            preds, vlog, _ = PO.pips_forward(self.pips_sd, xys[idx], fm, feat_init[idx], iters=iters)
            tr_o[:, idx] = preds[-1]
            vi_o[:, idx] = torch.sigmoid(vlog)
        return 0

    def sampt_pips_round_launches(self, h):
        return 0

    def sampt_pips_track_workspace_bytes(self, h, n, out):
        _set(out, 256)
        return 0

    def sampt_pips_track_f32(self, h, pyr, H0, W0, T, n, q, flip, q_host, flip_host, thr0, iters, evs, ev_lo, ev_hi, nchunks,
                             traj, vis, ws, nbytes, stream, rounds):
        """The device-side chain loop restated on the CPU (pips/tracker.py:42-153): window frames with the tail repeated,
        write-back of frames 1..hi-1, threshold linking in float32 — over ``sampt_pips_update_f32`` above."""
        S = 8
        assert tuple(q.shape) == (n, 3) and tuple(flip.shape) == (n,) and tuple(traj.shape) == (T, n, 2) and nchunks == 0
        traj.zero_(), vis.zero_()
        cur = q[:, 0].long().clone()
        ar = torch.arange(n)
        traj[cur, ar], vis[cur, ar] = q[:, 1:], 1.0
        feat_init = torch.zeros(n, 128)
        r = 0
        while True:
            act = (cur < T - 1).nonzero().flatten()
            if act.numel() == 0:
                break
            f = cur[act]
            hi = torch.minimum(T - f, torch.tensor(S))
            win = torch.minimum(f[:, None] + torch.arange(S)[None], (f + hi - 1)[:, None])
            used = torch.where(flip[act].bool()[:, None], T - 1 - win, win)
            xys = traj[f, act]
            if r == 0:
                out = torch.empty(act.numel(), 128)
                self.sampt_pips_sample_feat_f32(pyr[0], H0, W0, used[:, 0].int(), xys / 4.0, act.numel(), out, None)
                feat_init[act] = out
            tr_o, vi_o = torch.empty(S, act.numel(), 2), torch.empty(S, act.numel())
            self.sampt_pips_update_f32(h, pyr, H0, W0, used.int(), act.numel(), xys, feat_init[act], iters, tr_o, vi_o, None, 0, None)
            for j, i in enumerate(act.tolist()):
                fj, hj = int(f[j]), int(hi[j])
                traj[fj + 1:fj + hj, i], vis[fj + 1:fj + hj, i] = tr_o[1:hj, j], vi_o[1:hj, j]
                thr = torch.tensor(thr0, dtype=torch.float32)
                earliest, last = fj + 1, fj + hj - 1
                nxt = last
                while bool(vis[nxt, i] <= thr):
                    nxt -= 1
                    if nxt < earliest:
                        thr, nxt = thr - torch.tensor(0.02, dtype=torch.float32), last
                cur[i] = nxt
            r += 1
        _set(rounds, r)
        return 0

    # ------------------------------------------------------------------ SAM
    def sampt_vit_create(self, cfg_ref, names, ptrs, n, max_batch, out):
        self.calls["vit_create"] += 1
        c = cfg_ref._obj
        assert (c.embed_dim, c.depth, c.num_heads) == (self.cfg.embed_dim, self.cfg.depth, self.cfg.num_heads)
        return self._handle(out)

    def sampt_vit_destroy(self, h):
        pass

    def sampt_dec_create(self, names, ptrs, n, grid, img, max_frames, vit_dim, out):
        self.calls["dec_create"] += 1
        assert grid == self.cfg.grid and (vit_dim > 0) == self.hq
        return self._handle(out)

    def sampt_dec_destroy(self, h):
        pass

    def sampt_vit_encode_workspace_bytes(self, h, B, out):
        _set(out, 256)
        return 0

    def sampt_dec_workspace_bytes_k(self, h, frames, k, oh, ow, out):
        _set(out, 256)
        return 0

    def sampt_dec_hq_workspace_bytes(self, h, B, out):
        _set(out, 256)
        return 0

    def sampt_vit_live_rows(self, h, H, W, live, nbytes):
        # The skipping of frame-independent padding rows is an exact device-side optimisation.  By default the fake reports
        # nothing to skip; with ``model_live_rows`` set it follows the engine's rule (VitEngine::live_rows) so that the host
        # logic around sampt_vit_encode_live — one cache per frame geometry, built from the first frame — runs on the CPU.
        c = self.cfg
        lh = c.grid
        first_global = min(c.global_attn_indexes) if c.global_attn_indexes else c.depth
        if getattr(self, "model_live_rows", False) and first_global > 0 and W == c.img_size:
            h_tok = -(-H // c.patch_size)
            lh = min(c.grid, -(-h_tok // c.window_size) * c.window_size)
        live._obj.value = lh
        _set(nbytes, (c.grid - lh) * c.grid * c.embed_dim * 4)
        return 0

    def sampt_vit_encode_live(self, h, frames, chw, B, H, W, out, interm, cache, build, ws, nbytes, stream):
        if build:
            self.calls["vit_dead_cache_builds"] += 1
            assert B == 1 and out is None and cache is not None
            cache.fill_(float(H * 10000 + W))                                  # marker: which geometry the cache belongs to
            return 0
        assert float(cache[0]) == float(H * 10000 + W), "dead-row cache of another frame geometry"
        self.calls["vit_encode_live_frames"] += B
        return self.sampt_vit_encode(h, frames, chw, B, H, W, out, interm, ws, nbytes, stream)

    def sampt_vit_encode(self, h, frames, chw, B, H, W, out, interm, ws, nbytes, stream):
        from oracle import sam_ref as R
        self.calls["vit_encode_frames"] += B
        x = frames if chw else frames.permute(0, 3, 1, 2)
        assert x.dtype == torch.uint8 and tuple(x.shape) == (B, 3, H, W)
        x = R.preprocess(self.cfg, x.float())
        for i in range(B):
            if interm is not None:
                e, it = R.image_encoder(self.sam_sd, self.cfg, x[i:i + 1], return_interm=True)
                interm[i] = it[0].reshape(-1, it.shape[-1])
            else:
                e = R.image_encoder(self.sam_sd, self.cfg, x[i:i + 1])
            out[i] = e[0].permute(1, 2, 0).reshape(-1, e.shape[1])
        return 0

    def sampt_dec_hq_features(self, h, B, emb, interm, hq_out, ws, nbytes, stream):
        from oracle import sam_ref as R
        g = self.cfg.grid
        for i in range(B):
            e = emb[i].view(g, g, -1).permute(2, 0, 1)[None]
            hqf = R.hq_features(self.sam_sd, e, interm[i].view(1, g, g, -1))
            hq_out[i] = hqf[0].permute(1, 2, 0).reshape(-1, hqf.shape[1])
        return 0

    def _predict(self, feat, hq, pts, labels, k, box, mask, ih, iw, oh, ow):
        from oracle import sam_ref as R
        g = self.cfg.grid
        emb = feat.view(g, g, -1).permute(2, 0, 1)[None]
        sparse, dense = R.prompt_encoder(self.sam_sd, self.cfg, (pts[None, :k], labels[None, :k]),
                                         box.view(1, 4) if box is not None else None,
                                         mask.view(1, 1, 4 * g, 4 * g) if mask is not None else None)
        hq_feat = hq.view(4 * g, 4 * g, -1).permute(2, 0, 1)[None] if hq is not None else None
        low, iou = R.mask_decoder(self.sam_sd, self.cfg, emb, R.dense_pe(self.sam_sd, self.cfg), sparse, dense, False,
                                  hq_feat=hq_feat)
        return R.postprocess_masks(self.cfg, low, (ih, iw), (oh, ow)), iou, low

    def sampt_sam_decode(self, h, feat, hq, pts, labels, k, box, mask, ih, iw, oh, ow, logits, iou, low, ws, nbytes, stream):
        self.calls["sam_decode"] += 1
        m, i, l = self._predict(feat, hq, pts, labels, k, box, mask, ih, iw, oh, ow)
        logits.copy_(m.reshape(logits.shape)), iou.copy_(i.reshape(iou.shape)), low.copy_(l.reshape(low.shape))
        return 0

    def sampt_sam_track_decode(self, h, F, feats, hq, pts, labels, k, k_item, npos_item, ld, n_pos_first, R_, thr, ih, iw,
                               oh, ow, out_l, out_s, ws, nbytes, stream):
        self.calls["sam_track_decode"] += 1
        self.calls["sam_track_decode_items"] += F
        for f in range(F):
            kf = int(k_item[f]) if k_item is not None else k
            hqf = hq[f] if hq is not None else None
            low = None
            if n_pos_first >= 0:
                pf = int(npos_item[f]) if npos_item is not None else n_pos_first
                _, _, low = self._predict(feats[f], hqf, pts[f], labels[f], pf, None, None, ih, iw, oh, ow)
            m, iou, low = self._predict(feats[f], hqf, pts[f], labels[f], kf, None, low, ih, iw, oh, ow)
            for _ in range(R_):
                msk = m[0, 0] > 0
                if msk.sum() < 2:
                    break
                yx = msk.nonzero()
                box = torch.tensor([yx[:, 1].min(), yx[:, 0].min(), yx[:, 1].max(), yx[:, 0].max()], dtype=torch.float)
                m, iou, low = self._predict(feats[f], hqf, pts[f], labels[f], kf, box, low, ih, iw, oh, ow)
            out_s[f] = iou[0, 0]
            out_l[f] = m[0, 0] if float(iou[0, 0]) >= thr else -float("inf")
        return 0

    sampt_sam_track_decode_graph = sampt_sam_track_decode


def install(monkeypatch, sam_sd=None, cfg=None, pips_sd=None, hq=False) -> FakeHip:
    from sam_pt_amd import _lib
    fake = FakeHip(sam_sd, cfg, pips_sd, hq)
    monkeypatch.setattr(_lib, "load", lambda: fake)
    monkeypatch.setattr(_lib, "ptr", lambda t: t)
    monkeypatch.setattr(_lib, "ptr_array", lambda ts: list(ts))
    monkeypatch.setattr(_lib, "stream_ptr", lambda *a: None)
    monkeypatch.setattr(_lib, "name_table", lambda named: (named, None, len(named)))
    monkeypatch.setattr(_lib, "require_hip", lambda device, who: None)
    monkeypatch.setattr(_lib, "device_guard", lambda device: contextlib.nullcontext())
    return fake
