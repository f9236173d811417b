"""Support-set ensembling of the reweighting vectors (reference valid_ensemble.py:86-100): every class's vector is
the mean of meta_forward's outputs over ALL of its support shots, accumulated batch by batch as a running mean so
the support set never has to fit one batch.

    ens = ReweightEnsemble(n_cls)
    for metax, mask, clsids in metaloader:                 # dataset.MetaDataset(..., ensemble=True, with_ids=True)
        ens.add(model.meta_forward(metax, mask), clsids)
    dynamic_weights = ens.dynamic_weights()                # -> model.detect_forward(x, dynamic_weights)

The update is the reference's own expression, `e = e * cnt / (cnt + 1) + dw / (cnt + 1)`, evaluated on the device in
float32 in the same order, so the averaged vectors round exactly like the reference's.
"""
import torch


class ReweightEnsemble(object):
    def __init__(self, n_cls):
        self.n_cls = int(n_cls)
        self._mean = None            # list over reweighting layers of [n_cls] tensors (C, 1, 1), or 0.0 before the first shot
        self._cnt = [0.0] * self.n_cls

    def add(self, dws, clsids):
        """dws: meta_forward's return value (list of (n, C, 1, 1) tensors); clsids: the n class indices of the rows."""
        if isinstance(dws, torch.Tensor):
            dws = [dws]
        if self._mean is None:
            self._mean = [[0.0] * self.n_cls for _ in dws]
        if len(dws) != len(self._mean):
            raise ValueError("meta_forward returned %d vectors sets, the ensemble was started with %d" % (len(dws), len(self._mean)))
        ids = [int(c) for c in (clsids.tolist() if hasattr(clsids, "tolist") else clsids)]
        for dw in dws:
            if dw.shape[0] != len(ids):
                raise ValueError("%d reweighting vectors for %d class ids" % (dw.shape[0], len(ids)))
        for ci, c in enumerate(ids):
            if not 0 <= c < self.n_cls:
                raise ValueError("class id %d outside [0, %d)" % (c, self.n_cls))
            k = self._cnt[c]
            for layer, dw in enumerate(dws):
                self._mean[layer][c] = self._mean[layer][c] * k / (k + 1) + dw[ci].detach() / (k + 1)
            self._cnt[c] += 1

    @property
    def counts(self):
        return list(self._cnt)

    def dynamic_weights(self):
        """[ (n_cls, C, 1, 1) ] for Darknet.detect_forward; every class must have been seen at least once."""
        if self._mean is None or any(k == 0 for k in self._cnt):
            missing = [c for c, k in enumerate(self._cnt) if k == 0]
            raise ValueError("no support shot was added for classes %s" % missing)
        return [torch.stack(vs) for vs in self._mean]


def mean_by_group(dynamic_weights, group_sizes):
    """The other averaging mode of valid_ensemble.py:66-73: consecutive groups of rows (metaset.meta_cnts shots per
    class, all in one batch) replaced by their means."""
    out = []
    for dw in dynamic_weights:
        if dw.shape[0] != sum(group_sizes):
            raise ValueError("%d vectors for groups summing to %d" % (dw.shape[0], sum(group_sizes)))
        rows, lo = [], 0
        for n in group_sizes:
            rows.append(torch.mean(dw[lo:lo + n], dim=0))
            lo += n
        out.append(torch.stack(rows))
    return out
