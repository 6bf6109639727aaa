"""Host-side orchestration of the graph-VAE forward / backward over the HIP kernels.

One `Engine` owns views of the model's parameters / buffers (by reference state_dict name) and
issues the kernel sequence of `Encoder.forward`, `Decoder.forward` (reference model.py:466-483,
634-655) and their hand-written backward passes.  It contains no arithmetic of its own: every
tensor operation below is a C-ABI call (ops.py); torch only allocates device memory.

Data flow per GCL layer (model.py:55-121, 190-208):
    A    = segreduce(x)                [N,7d]   mean-aggregated messages of the 6 relations | x
    h    = A @ [W_0;..;W_5;root] + b   [N,d]    ONE fp32 MFMA GEMM, K = 7d (weight|root are adjacent
                                                in the flat parameter buffer)
    x'   = x + relu(BN(h))             [N,d]
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import constants as C
from . import ops
from ._lib import call, ptr, stream

F32 = torch.float32
EPS = 1e-5
MOM = 0.1
ENC_UID, DEC_UID = 0, 1000          # dropout-stream layer ids of the two GCNs
# stream ids of the `cfg.dropout` layers (element dropout, reference model.py line in the comment)
SITE = dict(enc_cnn_in=2001,     # CNNEncoder.lin[0]  :244
            enc_cnn_mid=2002,    # CNNEncoder.lin[3]  :247
            enc_chord=2003,      # ContentEncoder.dropout_layer on the chord embeddings :389-390
            enc_gate=2004,       # MLP.forward of the gate network :160
            enc_merge_in=2005,   # Encoder.dropout_layer :473
            enc_merge_out=2006,  # Encoder.dropout_layer :479
            dec_in=2007,         # Decoder.dropout :640
            dec_cnn_in=2008,     # CNNDecoder.lin[0]  :267
            dec_cnn_mid=2009,    # CNNDecoder.lin[3]  :270
            dec_chord=2010,      # ContentDecoder.dropout_layer :558-559
            enc_gcn=2100,        # GCN.forward :199, + layer index
            dec_gcn=2200)


class Engine:
    def __init__(self, cfg: dict, tensors: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.T = tensors
        self.d, self.nb, self.L = cfg["d"], cfg["n_bars"], cfg["gnn_n_layers"]
        self.msg_dropout = 0.1
        # synchronised BatchNorm (data parallel): a torch.distributed group, or None = per-replica statistics.
        # The reference is single-device: its statistics span what is here the GLOBAL batch.  With a group set, every
        # training-mode norm of this engine (GCL norms, CNN norms, heads, the gate's BatchNorm1d(1), the embedding norms)
        # sums its column sums over the ranks — one small all-reduce per norm and direction, one host read of the global
        # row counts per step.  Used by the DP-vs-single-device parity test; throughput runs keep per-replica statistics.
        self.sync_group = None
        self._sync_on = False
        self._rows = {}

    # ------------------------------------------------------------------ synchronised BatchNorm
    def set_sync_bn(self, group, on: bool = True):
        self.sync_group, self._sync_on = group, bool(on)

    def _reduce(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, group=self.sync_group)
        return t

    def _sync_rows(self, plan):
        """Global row counts of the three row spaces (nodes, bars, samples), keyed by the local count."""
        if plan.N in (plan.G, plan.B):
            raise NotImplementedError("synchronised BatchNorm: degenerate batch (one node per bar)")
        t = torch.tensor([plan.N, plan.G, plan.B], dtype=torch.float64, device=plan.buf.device)
        gN, gG, gB = self._reduce(t).tolist()                    # the one host read of the step
        self._rows = {plan.N: gN, plan.G: gG, plan.B: gB}

    def syncing(self, training) -> bool:
        return bool(training and self._sync_on)

    # ------------------------------------------------------------------ small helpers
    def p(self, name: str) -> torch.Tensor:
        return self.T[name]

    def _bump(self, key: str, training: bool, times: int = 1):
        if training:
            self.T[key + ".num_batches_tracked"] += times

    def lin(self, x, key, relu=False, out=None, M=None, lda=None, ldc=None):
        """y = x @ W^T + b for nn.Linear `key`; x may be a strided view (lda) and y a slice (ldc)."""
        W, b = self.T[key + ".weight"], self.T[key + ".bias"]
        N, K = W.shape
        M = x.shape[0] if M is None else M
        y = out if out is not None else torch.empty(M, N, dtype=F32, device=W.device)
        ops.gemm(x, W, y, M, N, K, lda or K, K, ldc or N, transB=True, bias=b, relu=relu)
        return y

    def lin_bwd(self, dy, x, key, G, M=None, ldx=None, lddy=None, dx_out=None, lddx=None, need_dx=True):
        """dW += dy^T x ; db += colsum(dy) ; dx = dy @ W (written to dx_out, leading dim lddx)."""
        W = self.T[key + ".weight"]
        N, K = W.shape
        M = dy.shape[0] if M is None else M
        lddy = lddy or N
        ops.gemm(dy, x, G[key + ".weight"], N, K, M, lddy, ldx or K, K, transA=True, accum=True, split_k=0)
        ops.colsum_acc(dy, M, N, lddy, G[key + ".bias"])
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else torch.empty(M, K, dtype=F32, device=W.device)
        ops.gemm(dy, W, dx, M, K, N, lddy, K, lddx or K)
        return dx

    def bn_train_or_eval(self, x, key, O, Cn, I, training, relu, residual=None):
        """BatchNorm (+ReLU, + residual) forward; returns (y, mean, var) — batch stats when training
        (running stats updated), running stats otherwise."""
        if self.syncing(training):
            sums = self._reduce(ops.bn_partial_sums(x, O, Cn, I))
            mean, var = ops.bn_stats_from_sums(sums, self._rows[O] * I, Cn, self.T[key + ".running_mean"],
                                               self.T[key + ".running_var"], MOM)
            self._bump(key, True)
        elif training:
            mean, var = ops.bn_stats(x, O, Cn, I, self.T[key + ".running_mean"], self.T[key + ".running_var"], MOM)
            self._bump(key, True)
        else:
            mean, var = self.T[key + ".running_mean"], self.T[key + ".running_var"]
        y = ops.bn_apply(x, O, Cn, I, mean, var, self.T[key + ".weight"], self.T[key + ".bias"], EPS, residual, relu)
        return y, mean, var

    def bn_back(self, x, dy, key, O, Cn, I, mean, var, relu, G, dbias_pre=None):
        if self._sync_on:
            ga, be = self.T[key + ".weight"], self.T[key + ".bias"]
            loc = ops.bn_partial_sums(x, O, Cn, I, dy=dy.contiguous(), mean=mean, var=var, gamma=ga, beta=be, eps=EPS, relu=relu)
            glob = self._reduce(loc.clone())
            return ops.bn_bwd_from_sums(x, dy.contiguous(), O, Cn, I, mean, var, ga, be, loc, glob, self._rows[O] * I,
                                        G[key + ".weight"], G[key + ".bias"], EPS, relu, dbias_pre=dbias_pre)
        return ops.bn_bwd(x, dy, O, Cn, I, mean, var, self.T[key + ".weight"], self.T[key + ".bias"],
                          G[key + ".weight"], G[key + ".bias"], EPS, relu, dbias_pre=dbias_pre)

    # ------------------------------------------------------------------ cfg.dropout (element dropout layers)
    def drop(self, x, cols, site, seed, training):
        """nn.Dropout(cfg.dropout) / F.dropout(p=cfg.dropout) of the reference at stream id `site`; identity in eval
        mode or with p = 0.  The backward of the layer is the same call on the gradient."""
        p = self.cfg["dropout"]
        if not training or not p:
            return x
        return ops.dropout_rows(x.contiguous(), cols, p, seed, site)

    def dropping(self, training) -> bool:
        return bool(training and self.cfg["dropout"])

    def _gcl_operand(self, lk: str) -> torch.Tensor:
        """[weight; root] of a GCL as ONE [7d, d] matrix: the two parameters are adjacent in the flat buffer."""
        W, R = self.T[lk + ".weight"], self.T[lk + ".root"]
        d = self.d
        if R.data_ptr() != W.data_ptr() + 4 * 6 * d * d:
            raise RuntimeError("GCL weight / root are not adjacent in the flat parameter buffer")
        return torch.as_strided(W, (7 * d, d), (d, 1))

    # ------------------------------------------------------------------ GCN (model.py:167-208)
    def gcn_forward(self, x, plan, key, training, seed, uid0):
        d, N = self.d, plan.N
        T = ops.edge_table(self.T[f"{key}.layers.0.nn.weight"], self.T[f"{key}.layers.0.nn.bias"])
        p = self.msg_dropout if training else 0.0
        site0 = SITE["enc_gcn" if uid0 == ENC_UID else "dec_gcn"]
        layers = []
        for i in range(self.L):
            lk = f"{key}.layers.{i}"
            xin = self.drop(x, d, site0 + i, seed, training)        # model.py:199 (the residual keeps the undropped x)
            A = ops.segreduce_fwd(xin, T, plan, p, seed, uid0 + i)
            if not training and self.cfg["batch_norm"]:
                # eval mode (generate.py:112): the norm is an affine map of running statistics, folded into the layer's
                # weights — x' = x + relu(A @ (W s) + (b s + t)) in ONE product whose epilogue adds the residual; the
                # normalisation pass over [N, d] is gone.  Folded per call: 7 d^2 floats per layer.
                nk = f"{key}.norm_layers.{i}.module"
                Wf, bf = ops.bn_fold_weights(self._gcl_operand(lk),
                                             self.T[lk + ".bias"], self.T[nk + ".weight"], self.T[nk + ".bias"],
                                             self.T[nk + ".running_mean"], self.T[nk + ".running_var"], EPS)
                xn = x.clone() if i == 0 else x          # (in place from layer 1 on: x is this loop's own buffer)
                ops.gemm(A, Wf, xn, N, d, 7 * d, 7 * d, d, d, bias=bf, relu_add=True)
                layers.append((xin, A, None, None, None))
                x = xn
                continue
            h = torch.empty(N, d, dtype=F32, device=x.device)
            # weight [6,d,d] and root [d,d] are adjacent in the flat buffer: B = [W_0;..;W_5;root], K = 7d
            ops.gemm(A, self.T[lk + ".weight"], h, N, d, 7 * d, 7 * d, d, d, bias=self.T[lk + ".bias"])
            if self.cfg["batch_norm"]:
                xn, mean, var = self.bn_train_or_eval(h, f"{key}.norm_layers.{i}.module", N, d, 1, training, True, x)
            else:                                                   # model.py:202-206 without the norm
                xn, mean, var = ops.relu_residual(h, x), None, None
            layers.append((xin, A, h, mean, var))
            x = xn
        return x, dict(T=T, layers=layers, seed=seed, uid0=uid0, p=p, site0=site0, drop=self.dropping(training))

    def gcn_backward(self, dx, sv, plan, key, G):
        d, N = self.d, plan.N
        T = sv["T"]
        dT = torch.zeros_like(T)
        for i in reversed(range(self.L)):
            lk = f"{key}.layers.{i}"
            x, A, h, mean, var = sv["layers"][i]
            if self.cfg["batch_norm"]:
                dh = self.bn_back(h, dx, f"{key}.norm_layers.{i}.module", N, d, 1, mean, var, True, G,
                                  dbias_pre=G[lk + ".bias"])      # GCL.bias sits right in front of the BatchNorm
            else:
                dh = ops.relu_bwd(dx, h)
                ops.colsum_acc(dh, N, d, d, G[lk + ".bias"])
            dA = torch.empty(N, 7 * d, dtype=F32, device=dx.device)
            ops.gemm(dh, self.T[lk + ".weight"], dA, N, 7 * d, d, d, d, 7 * d, transB=True)
            ops.gemm(A, dh, G[lk + ".weight"], 7 * d, d, N, 7 * d, d, d, transA=True, accum=True, split_k=0)
            if sv["drop"]:                 # x here is the DROPPED layer input; the residual branch bypasses the mask
                dxin = ops.segreduce_bwd(x, T, dA, None, plan, sv["p"], sv["seed"], sv["uid0"] + i, dT)
                dx = ops.add(dx, self.drop(dxin, d, sv["site0"] + i, sv["seed"], True))
            else:
                dx = ops.segreduce_bwd(x, T, dA, dx, plan, sv["p"], sv["seed"], sv["uid0"] + i, dT)
        ops.edge_table_bwd(dT, G[f"{key}.layers.0.nn.weight"], G[f"{key}.layers.0.nn.bias"])
        return dx

    # ------------------------------------------------------------------ encoder (model.py:420-483)
    def encoder_forward(self, plan, s_tensor, training, seed):
        d, nb, N, G_, B = self.d, self.nb, plan.N, plan.G, plan.B
        dev = s_tensor.device
        sv = {}
        if self.syncing(training):
            self._sync_rows(plan)
        zcat = torch.empty(B, 2 * d, dtype=F32, device=dev)          # [z_c | z_s]  (model.py:472)
        # --- structure encoder: CNN over the [G,1,4,32] grids
        k = "encoder.s_encoder.cnn_encoder"
        s = s_tensor.reshape(G_, 1, 4, 32)
        bn = self.cfg["batch_norm"]
        c4 = ".conv.4" if bn else ".conv.3"               # nn.Sequential index of the second conv (model.py:218-238)
        c0 = ops.conv3x3_fwd(s, self.T[k + ".conv.0.weight"], self.T[k + ".conv.0.bias"], G_, 1, 8, 4, 32)
        if bn:
            a0, m0, v0 = self.bn_train_or_eval(c0, k + ".conv.1", G_, 8, 128, training, True)
        else:
            a0, m0, v0 = ops.relu_residual(c0), None, None
        p0 = ops.maxpool4_fwd(a0)
        c1 = ops.conv3x3_fwd(p0, self.T[k + c4 + ".weight"], self.T[k + c4 + ".bias"], G_, 8, 16, 4, 8)
        if bn:
            a1, m1, v1 = self.bn_train_or_eval(c1, k + ".conv.5", G_, 16, 32, training, True)
        else:
            a1, m1, v1 = ops.relu_residual(c1), None, None
        a1d = self.drop(a1.view(G_, 512), 512, SITE["enc_cnn_in"], seed, training)
        h1 = self.lin(a1d, k + ".lin.1", relu=True)
        h1d = self.drop(h1, d, SITE["enc_cnn_mid"], seed, training)
        h2 = self.lin(h1d, k + ".lin.4")
        self.lin(h2, "encoder.s_encoder.bars_encoder", out=zcat[:, d:], M=B, lda=nb * d, ldc=2 * d)
        sv.update(s=s, c0=c0, a0=a0, m0=m0, v0=v0, p0=p0, c1=c1, a1=a1, a1d=a1d, m1=m1, v1=v1, h1=h1, h1d=h1d, h2=h2,
                  seed=seed)
        # --- content encoder: token embeddings -> chord embedding -> GCN -> attention pool
        k = "encoder.c_encoder"
        dh = d // 2
        tables = torch.empty(4, C.N_PITCH_TOKENS, dh, dtype=F32, device=dev)
        stats = torch.empty(4, 2, dh, dtype=F32, device=dev)
        Tn = self.T
        hist = plan.tok_hist
        if self.syncing(training):                       # the embedding norms see the token histogram of ALL ranks
            hist = self._reduce(plan.tok_hist.clone())
        call("pm_embed_tables", ptr(Tn[k + ".drums_pitch_emb.weight"]), ptr(Tn[k + ".drums_pitch_emb.bias"]),
             ptr(Tn[k + ".non_drums_pitch_emb.weight"]), ptr(Tn[k + ".non_drums_pitch_emb.bias"]),
             ptr(Tn[k + ".dur_emb.weight"]), ptr(Tn[k + ".dur_emb.bias"]),
             ptr(Tn[k + ".bn_drums.weight"]), ptr(Tn[k + ".bn_drums.bias"]),
             ptr(Tn[k + ".bn_non_drums.weight"]), ptr(Tn[k + ".bn_non_drums.bias"]),
             ptr(Tn[k + ".bn_dur.weight"]), ptr(Tn[k + ".bn_dur.bias"]),
             ptr(Tn[k + ".bn_drums.running_mean"]), ptr(Tn[k + ".bn_drums.running_var"]),
             ptr(Tn[k + ".bn_non_drums.running_mean"]), ptr(Tn[k + ".bn_non_drums.running_var"]),
             ptr(Tn[k + ".bn_dur.running_mean"]), ptr(Tn[k + ".bn_dur.running_var"]),
             ptr(hist), d, int(training), EPS, MOM, ptr(tables), ptr(stats), stream())
        if training:
            # one update per non-empty group; bn_dur is applied to both groups (model.py:362,375)
            cnt = plan.group_cnt[:2].to(torch.int64)
            if self.syncing(training):
                cnt = hist.view(4, -1)[:2].sum(dim=1).to(torch.int64)
            has = (cnt > 0).to(torch.int64)
            Tn[k + ".bn_drums.num_batches_tracked"] += has[0]
            Tn[k + ".bn_non_drums.num_batches_tracked"] += has[1]
            Tn[k + ".bn_dur.num_batches_tracked"] += has[0] + has[1]
        X = torch.empty(N, C.N_SLOTS * d, dtype=F32, device=dev)
        call("pm_embed_gather", ptr(tables), ptr(plan.tokens), ptr(plan.is_drum), N, d, C.N_SLOTS, ptr(X), stream())
        x0 = self.lin(X, k + ".chord_encoder", relu=True)
        x0d = self.drop(x0, d, SITE["enc_chord"], seed, training)                 # model.py:389-390 (row = node)
        xL, gsv = self.gcn_forward(x0d, plan, k + ".graph_encoder", training, seed, ENC_UID)
        gk = k + ".graph_attention.gate_nn"
        xLg = self.drop(xL, d, SITE["enc_gate"], seed, training)                  # MLP.forward, model.py:160
        g = ops.gate_fwd(xLg, Tn[gk + ".0.layers.0.weight"].view(-1), Tn[gk + ".0.layers.0.bias"])
        if self.syncing(training):
            gs = self._reduce(ops.bn_partial_sums(g, N, 1, 1))
            gm, gv = ops.bn_stats_from_sums(gs, self._rows[N], 1, Tn[gk + ".1.running_mean"], Tn[gk + ".1.running_var"], MOM)
            self._bump(gk + ".1", True)
        elif training:
            gm, gv = ops.bn_stats(g, N, 1, 1, Tn[gk + ".1.running_mean"], Tn[gk + ".1.running_var"], MOM)
            self._bump(gk + ".1", True)
        else:
            gm, gv = Tn[gk + ".1.running_mean"], Tn[gk + ".1.running_var"]
        alpha, pooled = ops.attnpool_fwd(xL, g, gm, gv, Tn[gk + ".1.weight"], Tn[gk + ".1.bias"], plan, EPS)
        self.lin(pooled, k + ".bars_encoder", out=zcat, M=B, lda=nb * d, ldc=2 * d)
        sv.update(stats=stats, X=X, x0=x0, gcn=gsv, xL=xL, xLg=xLg, g=g, gm=gm, gv=gv, alpha=alpha, pooled=pooled, hist=hist)
        # --- merge + heads (model.py:472-481)
        zcat_d = self.drop(zcat, 2 * d, SITE["enc_merge_in"], seed, training)
        m = self.lin(zcat_d, "encoder.linear_merge")
        zg0, mm, mv = self.bn_train_or_eval(m, "encoder.bn_linear_merge", B, d, 1, training, True)
        zg = self.drop(zg0, d, SITE["enc_merge_out"], seed, training)
        mu = self.lin(zg, "encoder.linear_mu")
        lv = self.lin(zg, "encoder.linear_log_var")
        sv.update(zcat=zcat_d, m=m, mm=mm, mv=mv, zg=zg, mu=mu, plan=plan, training=training)
        return mu, lv, sv

    def encoder_backward(self, sv, dmu, dlv, G):
        if not sv["training"]:
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented on the HIP path")
        d, nb = self.d, self.nb
        plan = sv["plan"]
        N, G_, B = plan.N, plan.G, plan.B
        Tn = self.T
        dev = dmu.device
        # heads
        dzg = self.lin_bwd(dmu, sv["zg"], "encoder.linear_mu", G)
        dzg2 = self.lin_bwd(dlv, sv["zg"], "encoder.linear_log_var", G)
        dzg = ops.add(dzg, dzg2)
        seed, tr = sv["seed"], True
        dzg = self.drop(dzg, d, SITE["enc_merge_out"], seed, tr)
        dm = self.bn_back(sv["m"], dzg, "encoder.bn_linear_merge", B, d, 1, sv["mm"], sv["mv"], True, G)
        dzcat = self.lin_bwd(dm, sv["zcat"], "encoder.linear_merge", G)
        dzcat = self.drop(dzcat, 2 * d, SITE["enc_merge_in"], seed, tr)
        # --- content branch: z_c = zcat[:, :d]
        k = "encoder.c_encoder"
        dpooled = self.lin_bwd(dzcat, sv["pooled"], k + ".bars_encoder", G, M=B, ldx=nb * d, lddy=2 * d)
        gk = k + ".graph_attention.gate_nn"
        pool_args = (sv["xL"], sv["g"], sv["gm"], sv["gv"], Tn[gk + ".1.weight"], sv["alpha"], dpooled.view(G_, d),
                     Tn[gk + ".0.layers.0.weight"].view(-1), plan, G[gk + ".0.layers.0.weight"].view(-1),
                     G[gk + ".0.layers.0.bias"], G[gk + ".1.weight"], G[gk + ".1.bias"])
        xg = sv["xLg"] if self.dropping(True) else None
        if self._sync_on:
            dxL = ops.attnpool_bwd_sync(*pool_args, self._reduce, self._rows[N], EPS, x_gate=xg)
        else:
            dxL = ops.attnpool_bwd(*pool_args, EPS, x_gate=xg)
        if self.dropping(True):            # pooled path + masked gate path
            dxL = ops.add(dxL[0], self.drop(dxL[1], d, SITE["enc_gate"], seed, tr))
        dx0 = self.gcn_backward(dxL, sv["gcn"], plan, k + ".graph_encoder", G)
        dx0 = self.drop(dx0, d, SITE["enc_chord"], seed, tr)
        dx0 = ops.relu_bwd(dx0, sv["x0"])
        dX = self.lin_bwd(dx0, sv["X"], k + ".chord_encoder", G)
        dh = d // 2
        S = torch.empty(4, C.N_PITCH_TOKENS, dh, dtype=F32, device=dev)
        call("pm_embed_bwd_scatter", ptr(dX), ptr(plan.tokens), ptr(plan.buf), N, plan.E, G_, d, C.N_SLOTS, ptr(S), stream())
        emb_args = (ptr(S), ptr(Tn[k + ".drums_pitch_emb.weight"]), ptr(Tn[k + ".drums_pitch_emb.bias"]),
                    ptr(Tn[k + ".non_drums_pitch_emb.weight"]), ptr(Tn[k + ".non_drums_pitch_emb.bias"]),
                    ptr(Tn[k + ".dur_emb.weight"]), ptr(Tn[k + ".dur_emb.bias"]), ptr(Tn[k + ".bn_drums.weight"]),
                    ptr(Tn[k + ".bn_non_drums.weight"]), ptr(Tn[k + ".bn_dur.weight"]), ptr(sv["stats"]), ptr(plan.tok_hist),
                    d, EPS, ptr(G[k + ".drums_pitch_emb.weight"]), ptr(G[k + ".drums_pitch_emb.bias"]),
                    ptr(G[k + ".non_drums_pitch_emb.weight"]), ptr(G[k + ".non_drums_pitch_emb.bias"]),
                    ptr(G[k + ".dur_emb.weight"]), ptr(G[k + ".dur_emb.bias"]), ptr(G[k + ".bn_drums.weight"]),
                    ptr(G[k + ".bn_drums.bias"]), ptr(G[k + ".bn_non_drums.weight"]), ptr(G[k + ".bn_non_drums.bias"]),
                    ptr(G[k + ".bn_dur.weight"]), ptr(G[k + ".bn_dur.bias"]))
        if self._sync_on:        # the two batch means of the embedding norms' backward over all ranks (two phases)
            es = torch.empty(4, 2, dh, dtype=torch.float64, device=dev)
            call("pm_embed_tables_bwd_sync", *emb_args, ptr(es), None, None, stream())
            self._reduce(es)
            call("pm_embed_tables_bwd_sync", *emb_args, None, ptr(es), ptr(sv["hist"]), stream())
        else:
            call("pm_embed_tables_bwd", *emb_args, stream())
        # --- structure branch: z_s = zcat[:, d:]
        k = "encoder.s_encoder.cnn_encoder"
        dh2 = self.lin_bwd(dzcat[:, d:], sv["h2"], "encoder.s_encoder.bars_encoder", G, M=B, ldx=nb * d, lddy=2 * d)
        bn = self.cfg["batch_norm"]
        c4 = ".conv.4" if bn else ".conv.3"
        dh1 = self.lin_bwd(dh2.view(G_, d), sv["h1d"], k + ".lin.4", G)
        dh1 = self.drop(dh1, d, SITE["enc_cnn_mid"], seed, tr)
        dh1 = ops.relu_bwd(dh1, sv["h1"])
        da1 = self.lin_bwd(dh1, sv["a1d"], k + ".lin.1", G)
        da1 = self.drop(da1, 512, SITE["enc_cnn_in"], seed, tr)
        if bn:
            dc1 = self.bn_back(sv["c1"], da1.view(G_, 16, 4, 8), k + ".conv.5", G_, 16, 32, sv["m1"], sv["v1"], True, G)
        else:
            dc1 = ops.relu_bwd(da1.view(G_, 16, 4, 8), sv["c1"])
        ops.conv3x3_bwd_weight(sv["p0"], dc1, G_, 8, 16, 4, 8, G[k + c4 + ".weight"], G[k + c4 + ".bias"])
        dp0 = ops.conv3x3_bwd_data(dc1, Tn[k + c4 + ".weight"], G_, 8, 16, 4, 8)
        da0 = ops.maxpool4_bwd(sv["a0"], dp0)
        if bn:
            dc0 = self.bn_back(sv["c0"], da0, k + ".conv.1", G_, 8, 128, sv["m0"], sv["v0"], True, G)
        else:
            dc0 = ops.relu_bwd(da0, sv["c0"])
        ops.conv3x3_bwd_weight(sv["s"], dc0, G_, 1, 8, 4, 32, G[k + ".conv.0.weight"], G[k + ".conv.0.bias"])

    # ------------------------------------------------------------------ decoder (model.py:486-655)
    def _structure_decoder(self, zr, B, G_, training, update_stats=True, seed=0):
        d, nb = self.d, self.nb
        k = "decoder.s_decoder"
        sb = self.lin(zr, k + ".bars_decoder", M=B, lda=2 * d)                       # A = zr[:, :d]
        drop = training and update_stats            # (the structure-only pass of the generation path runs in eval mode)
        sbd = self.drop(sb.view(G_, d), d, SITE["dec_cnn_in"], seed, drop)
        u1 = self.lin(sbd, k + ".cnn_decoder.lin.1", relu=True)
        u1d = self.drop(u1, d, SITE["dec_cnn_mid"], seed, drop)
        u2 = self.lin(u1d, k + ".cnn_decoder.lin.4", relu=True)                     # [G,512] = [G,16,4,8]
        ck = k + ".cnn_decoder.conv"
        c2 = ops.conv3x3_fwd(u2, self.T[ck + ".1.weight"], self.T[ck + ".1.bias"], G_, 16, 8, 4, 32, up4=True)
        if not self.cfg["batch_norm"]:                                               # model.py:278-292 without BatchNorm2d
            a2 = ops.relu_residual(c2)
            s_logits = ops.conv3x3_fwd(a2, self.T[ck + ".3.weight"], self.T[ck + ".3.bias"], G_, 8, 1, 4, 32)
            return s_logits.view(B, nb, 4, 32), dict(sb=sb, sbd=sbd, u1=u1, u1d=u1d, u2=u2, c2=c2, a2=a2, m2=None, v2=None)
        if training and update_stats:
            a2, m2, v2 = self.bn_train_or_eval(c2, ck + ".2", G_, 8, 128, True, True)
        elif training:
            m2, v2 = ops.bn_stats(c2, G_, 8, 128)
            a2 = ops.bn_apply(c2, G_, 8, 128, m2, v2, self.T[ck + ".2.weight"], self.T[ck + ".2.bias"], EPS, None, True)
        else:
            a2, m2, v2 = self.bn_train_or_eval(c2, ck + ".2", G_, 8, 128, False, True)
        s_logits = ops.conv3x3_fwd(a2, self.T[ck + ".4.weight"], self.T[ck + ".4.bias"], G_, 8, 1, 4, 32)
        return s_logits.view(B, nb, 4, 32), dict(sb=sb, sbd=sbd, u1=u1, u1d=u1d, u2=u2, c2=c2, a2=a2, m2=m2, v2=v2)

    def structure_only(self, z, training):
        """s_logits of `Decoder.forward` without touching running statistics (generation path: the
        structure must exist before the content decoder can run, model.py:646-650)."""
        B, d = z.shape[0], self.d
        zd = self.lin(z, "decoder.lin_decoder")
        if training:
            dm, dv = ops.bn_stats(zd, B, 2 * d, 1)
        else:
            dm, dv = self.T["decoder.batch_norm.running_mean"], self.T["decoder.batch_norm.running_var"]
        zr = ops.bn_apply(zd, B, 2 * d, 1, dm, dv, self.T["decoder.batch_norm.weight"],
                          self.T["decoder.batch_norm.bias"], EPS, None, True)
        s_logits, _ = self._structure_decoder(zr, B, B * self.nb, training, update_stats=False)
        return s_logits

    def decoder_forward(self, plan, z, training, seed):
        d, nb, N, G_, B = self.d, self.nb, plan.N, plan.G, plan.B
        dev = z.device
        Tn = self.T
        if self.syncing(training) and plan.N not in self._rows:
            self._sync_rows(plan)
        zd = self.lin(z, "decoder.lin_decoder")
        zr0, dm, dv = self.bn_train_or_eval(zd, "decoder.batch_norm", B, 2 * d, 1, training, True)
        zr = self.drop(zr0, 2 * d, SITE["dec_in"], seed, training)                  # model.py:640
        s_logits, ssv = self._structure_decoder(zr, B, G_, training, seed=seed)
        # content decoder
        k = "decoder.c_decoder"
        cb = self.lin(zr[:, d:], k + ".bars_decoder", M=B, lda=2 * d)               # A = zr[:, d:]
        x0 = ops.bar_broadcast_fwd(cb.view(G_, d), plan)
        xL, gsv = self.gcn_forward(x0, plan, k + ".graph_decoder", training, seed, DEC_UID)
        H = self.lin(xL, k + ".chord_decoder")                                      # [N, 15*d]
        H = self.drop(H, C.N_SLOTS * d, SITE["dec_chord"], seed, training)          # model.py:558-559 (row = node)
        c_logits = torch.empty(N, C.N_SLOTS, C.D_TOKEN_PAIR, dtype=F32, device=dev)
        R, dh = N * C.N_SLOTS, d // 2
        Hf, Lf = H.view(-1), c_logits.view(-1)
        # duration logits for every row; pitch logits per drum / non-drum node list (model.py:561-576)
        ops.gemm(Hf[dh:], Tn[k + ".dur_emb.weight"], Lf[C.N_PITCH_TOKENS:], R, C.N_DUR_TOKENS, dh, d, dh,
                 C.D_TOKEN_PAIR, transB=True, bias=Tn[k + ".dur_emb.bias"])
        cnt = plan.group_cnt
        for name, lst, c in ((".drums_pitch_emb", plan.drum_list, cnt[0:1]), (".non_drums_pitch_emb", plan.nondrum_list, cnt[1:2])):
            ops.gemm(Hf, Tn[k + name + ".weight"], Lf, R, C.N_PITCH_TOKENS, dh, d, dh, C.D_TOKEN_PAIR, transB=True,
                     bias=Tn[k + name + ".bias"], rowmap=lst, rows_per_entry=C.N_SLOTS, dyn_entries=c)
        sv = dict(z=z, zd=zd, dm=dm, dv=dv, zr=zr, s=ssv, cb=cb, x0=x0, gcn=gsv, xL=xL, H=H, plan=plan,
                  training=training, seed=seed)
        return s_logits, c_logits, sv

    def decoder_backward(self, sv, ds_logits, dc_logits, G):
        if not sv["training"]:
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented on the HIP path")
        d, nb = self.d, self.nb
        plan = sv["plan"]
        N, G_, B = plan.N, plan.G, plan.B
        Tn = self.T
        dev = sv["z"].device
        dzr = torch.zeros(B, 2 * d, dtype=F32, device=dev)
        seed, tr = sv["seed"], True
        k = "decoder.c_decoder"
        if dc_logits is not None:
            R, dh = N * C.N_SLOTS, d // 2
            H = sv["H"]
            dH = torch.empty_like(H)
            Hf, dHf, dLf = H.view(-1), dH.view(-1), dc_logits.view(-1)
            NP, ND, NT = C.N_PITCH_TOKENS, C.N_DUR_TOKENS, C.D_TOKEN_PAIR
            # duration un-embedding
            ops.gemm(dLf[NP:], Tn[k + ".dur_emb.weight"], dHf[dh:], R, dh, ND, NT, dh, d)
            ops.gemm(dLf[NP:], Hf[dh:], G[k + ".dur_emb.weight"], ND, dh, R, NT, d, dh, transA=True, accum=True, split_k=0)
            ops.colsum_acc(dLf[NP:], R, ND, NT, G[k + ".dur_emb.bias"])
            cnt = plan.group_cnt
            for name, lst, c in ((".drums_pitch_emb", plan.drum_list, cnt[0:1]),
                                 (".non_drums_pitch_emb", plan.nondrum_list, cnt[1:2])):
                ops.gemm(dLf, Tn[k + name + ".weight"], dHf, R, dh, NP, NT, dh, d, rowmap=lst,
                         rows_per_entry=C.N_SLOTS, dyn_entries=c)
                ops.gemm(dLf, Hf, G[k + name + ".weight"], NP, dh, R, NT, d, dh, transA=True, accum=True, split_k=0,
                         rowmap=lst, rows_per_entry=C.N_SLOTS, dyn_entries=c)
                ops.colsum_rows_acc(dLf, NP, NT, lst, C.N_SLOTS, c, N, G[k + name + ".bias"])
            dH = self.drop(dH, C.N_SLOTS * d, SITE["dec_chord"], seed, tr)
            dxL = self.lin_bwd(dH, sv["xL"], k + ".chord_decoder", G)
            dx0 = self.gcn_backward(dxL, sv["gcn"], plan, k + ".graph_decoder", G)
            dcb = ops.bar_broadcast_bwd(dx0, plan)
            self.lin_bwd(dcb, sv["zr"][:, d:], k + ".bars_decoder", G, M=B, ldx=2 * d, dx_out=dzr[:, d:], lddx=2 * d)
        if ds_logits is not None:
            k = "decoder.s_decoder"
            ck = k + ".cnn_decoder.conv"
            s = sv["s"]
            dsl = ds_logits.reshape(G_, 1, 4, 32).contiguous()
            cl = ".4" if self.cfg["batch_norm"] else ".3"
            ops.conv3x3_bwd_weight(s["a2"], dsl, G_, 8, 1, 4, 32, G[ck + cl + ".weight"], G[ck + cl + ".bias"])
            da2 = ops.conv3x3_bwd_data(dsl, Tn[ck + cl + ".weight"], G_, 8, 1, 4, 32)
            if self.cfg["batch_norm"]:
                dc2 = self.bn_back(s["c2"], da2, ck + ".2", G_, 8, 128, s["m2"], s["v2"], True, G)
            else:
                dc2 = ops.relu_bwd(da2, s["c2"])
            ops.conv3x3_bwd_weight(s["u2"], dc2, G_, 16, 8, 4, 32, G[ck + ".1.weight"], G[ck + ".1.bias"], up4=True)
            du2 = ops.conv3x3_bwd_data(dc2, Tn[ck + ".1.weight"], G_, 16, 8, 4, 32, up4=True)
            du2 = ops.relu_bwd(du2.view(G_, 512), s["u2"])
            du1 = self.lin_bwd(du2, s["u1d"], k + ".cnn_decoder.lin.4", G)
            du1 = self.drop(du1, d, SITE["dec_cnn_mid"], seed, tr)
            du1 = ops.relu_bwd(du1, s["u1"])
            dsb = self.lin_bwd(du1, s["sbd"], k + ".cnn_decoder.lin.1", G)
            dsb = self.drop(dsb, d, SITE["dec_cnn_in"], seed, tr)
            self.lin_bwd(dsb.view(B, nb * d), sv["zr"], k + ".bars_decoder", G, M=B, ldx=2 * d, dx_out=dzr, lddx=2 * d)
        dzr = self.drop(dzr, 2 * d, SITE["dec_in"], seed, tr)
        dzd = self.bn_back(sv["zd"], dzr, "decoder.batch_norm", B, 2 * d, 1, sv["dm"], sv["dv"], True, G)
        dz = self.lin_bwd(dzd, sv["z"], "decoder.lin_decoder", G)
        return dz
