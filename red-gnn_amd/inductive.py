"""DataLoader of the inductive setting with the reference's interface (Static/inductive/load_data.py:7-192).

Two graphs: the training graph over ``task_dir`` (``mode='transductive'``) and the test graph over
``task_dir + '_ind'`` with its own entity set (``mode='inductive'``); relations are shared.  Same quirks as the
reference, kept on purpose: every triple is stored doubled ([h,r,t] then [t,r+n_rel,h], load_data.py:87-88), the
*training* queries are the triples of ``valid.txt`` while ``train.txt`` builds the graph (load_data.py:57-60),
validation queries come from ``test.txt`` and test queries from the ``_ind`` valid + test files (load_data.py:61-65).
"""
import os
from collections import defaultdict

import numpy as np
import torch

from .engine import Frontier, Graph, _require_gpu


class DataLoader:
    inductive = True

    def __init__(self, task_dir=None, ids=None, device="cuda", verbose=True):
        self.device = torch.device(device)
        if ids is None:
            self._read_text(task_dir)
        else:
            self.n_ent, self.n_rel, self.n_ent_ind = int(ids["n_ent"]), int(ids["n_rel"]), int(ids["n_ent_ind"])
            as_l = lambda a: np.asarray(a, dtype=np.int64).reshape(-1, 3)
            self.tra_train_all, self.tra_valid, self.tra_test = as_l(ids["tra_kg"]), as_l(ids["tra_valid"]), as_l(ids["tra_test"])
            self.ind_train, self.ind_valid, self.ind_test = as_l(ids["ind_kg"]), as_l(ids["ind_valid"]), as_l(ids["ind_test"])
        self.val_filters = self.get_filter("valid")
        self.tst_filters = self.get_filter("test")
        self._graphs = {}
        self.tra_train = self.tra_valid.copy()                                   # load_data.py:60
        self.valid_q, self.valid_a = self.load_query(self.tra_test)              # load_data.py:61,64
        ivq, iva = self.load_query(self.ind_valid)
        itq, ita = self.load_query(self.ind_test)
        self.test_q, self.test_a = ivq + itq, iva + ita                          # load_data.py:65
        self.n_train, self.n_valid, self.n_test = len(self.tra_train), len(self.valid_q), len(self.test_q)
        self._frontiers = {}
        if verbose:
            print("n_train:", self.n_train, "n_valid:", self.n_valid, "n_test:", self.n_test)

    # ---- parsing (load_data.py:12-47, 77-90): "name id" pairs per line -----------------------------------
    def _read_text(self, task_dir):
        ind_dir = task_dir + "_ind"

        def pairs(path):
            with open(path) as f:
                return {k: int(v) for k, v in (line.strip().split() for line in f)}
        self.entity2id = pairs(os.path.join(task_dir, "entities.txt"))
        self.relation2id = pairs(os.path.join(task_dir, "relations.txt"))
        self.entity2id_ind = pairs(os.path.join(ind_dir, "entities.txt"))
        self.n_ent, self.n_rel, self.n_ent_ind = len(self.entity2id), len(self.relation2id), len(self.entity2id_ind)

        def read(directory, filename, e2i):
            rows = []
            with open(os.path.join(directory, filename)) as f:
                for line in f:
                    h, r, t = line.strip().split()
                    h, r, t = e2i[h], self.relation2id[r], e2i[t]
                    rows.append((h, r, t))
                    rows.append((t, r + self.n_rel, h))
            return np.array(rows, dtype=np.int64).reshape(-1, 3)
        self.tra_train_all = read(task_dir, "train.txt", self.entity2id)
        self.tra_valid = read(task_dir, "valid.txt", self.entity2id)
        self.tra_test = read(task_dir, "test.txt", self.entity2id)
        self.ind_train = read(ind_dir, "train.txt", self.entity2id_ind)
        self.ind_valid = read(ind_dir, "valid.txt", self.entity2id_ind)
        self.ind_test = read(ind_dir, "test.txt", self.entity2id_ind)

    def get_filter(self, data="valid"):
        """load_data.py:162-192."""
        filters = defaultdict(set)
        parts = (self.tra_train_all, self.tra_valid, self.tra_test) if data == "valid" else (self.ind_train, self.ind_valid, self.ind_test)
        for part in parts:
            for h, r, t in part.tolist():
                filters[(h, r)].add(t)
        return {k: sorted(v) for k, v in filters.items()}

    def load_query(self, triples):
        by_hr = defaultdict(list)
        for h, r, t in sorted(map(tuple, np.asarray(triples).tolist()), key=lambda x: (x[0], x[1])):
            by_hr[(h, r)].append(t)
        queries = list(by_hr.keys())
        return queries, [np.array(by_hr[k]) for k in queries]

    # ---- graphs (load_data.py:92-103): triples are already doubled, identity rows come from rg_graph_create ----
    def graph_for(self, mode):
        mode = "transductive" if mode in ("transductive", "train") else "inductive"
        g = self._graphs.get(mode)
        if g is None:
            trip, n_ent = (self.tra_train_all, self.n_ent) if mode == "transductive" else (self.ind_train, self.n_ent_ind)
            g = self._graphs[mode] = Graph(n_ent, self.n_rel, trip, add_inverse=False, device=self.device)
        return g

    def eval_mode(self, data):
        """base_model.py of the inductive setting: validation on the training graph, test on the inductive one."""
        return "transductive" if data == "valid" else "inductive"

    def get_neighbors(self, nodes, mode="transductive"):
        """load_data.py:120-146 — same contract as the transductive loader's."""
        _require_gpu(self.device)
        nodes_t = torch.as_tensor(np.asarray(nodes) if not torch.is_tensor(nodes) else nodes)
        nodes_t = nodes_t.to(device=self.device, dtype=torch.int32).contiguous()
        n_batch = int(nodes_t[:, 0].max().item()) + 1 if nodes_t.numel() else 1
        graph = self.graph_for(mode)
        key = (graph.n_ent, n_batch)
        fr = self._frontiers.get(key)
        if fr is None:
            if len(self._frontiers) > 8:           # a handful of batch sizes is the norm; do not hoard workspaces
                self._frontiers.clear()
            fr = self._frontiers[key] = Frontier(graph.n_ent, n_batch, 2, self.device)
        fr.reset_nodes(nodes_t)
        fr.expand(graph)
        tail_nodes, _, old_new = fr.nodes(want_prev=False)
        edges, _ = fr.edges(graph, tail_nodes)
        return tail_nodes.long(), edges.long(), old_new.long()

    def get_batch(self, batch_idx, steps=2, data="train"):
        """load_data.py:148-166."""
        if data == "train":
            return self.tra_train[batch_idx]
        query, answer = (self.valid_q, self.valid_a) if data == "valid" else (self.test_q, self.test_a)
        n_ent = self.n_ent if data == "valid" else self.n_ent_ind
        batch_idx = np.asarray(batch_idx)
        subs = np.array([query[i][0] for i in batch_idx])
        rels = np.array([query[i][1] for i in batch_idx])
        objs = np.zeros((len(batch_idx), n_ent))
        for i, q in enumerate(batch_idx):
            objs[i][answer[q]] = 1
        return subs, rels, objs

    def get_batch_csr(self, batch_idx, data="valid", device_queries=False):
        """As load_data.DataLoader.get_batch_csr (subs / rels stay numpy arrays here: the inductive splits are small)."""
        query, answer = (self.valid_q, self.valid_a) if data == "valid" else (self.test_q, self.test_a)
        filters = self.val_filters if data == "valid" else self.tst_filters
        subs = np.array([query[i][0] for i in batch_idx])
        rels = np.array([query[i][1] for i in batch_idx])
        ans = [np.sort(np.asarray(answer[i])) for i in batch_idx]
        fil = [np.asarray(filters[(int(s), int(r))]) for s, r in zip(subs, rels)]
        to_dev = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(self.device)
        ptr = lambda lists: np.concatenate([[0], np.cumsum([len(x) for x in lists])])
        cat = lambda lists: np.concatenate(lists) if len(lists) else np.zeros(0, np.int64)
        return subs, rels, to_dev(ptr(ans)), to_dev(cat(ans)), to_dev(ptr(fil)), to_dev(cat(fil))

    def shuffle_train(self):
        """load_data.py:168-170."""
        self.tra_train = self.tra_train[np.random.permutation(self.n_train)]
