"""DataLoader with the reference's interface (Static/transductive/load_data.py:7-164), backed by
device-resident graphs and the HIP frontier expansion.

Same constructor argument (``task_dir`` with entities.txt / relations.txt / facts.txt / train.txt /
valid.txt / test.txt), same attributes the trainer reads (``n_ent, n_rel, n_train, n_valid, n_test,
filters, valid_q/valid_a, test_q/test_a, train_data``) and the same methods (``get_neighbors``,
``get_batch``, ``shuffle_train``).  Differences, all internal: the KG is int32 on the device instead
of a float64 numpy array + scipy CSR (load_data.py:77-81), and ``get_neighbors`` runs on the GPU.
``ids=`` builds the loader from pre-parsed id arrays (fixtures, synthetic KGs) instead of text.
"""
import os
from collections import defaultdict

import numpy as np
import torch

from .engine import Frontier, Graph, _require_gpu


class DataLoader:
    def __init__(self, task_dir=None, ids=None, device="cuda", verbose=True, cache_dir=None):
        self.task_dir = task_dir
        self.device = torch.device(device)
        self._filters = defaultdict(set)
        binary = None
        if ids is None and cache_dir is not None:
            ids = self._cached_ids(task_dir, cache_dir)      # binary cache of the parsed text
        if ids is None:
            self._read_text(task_dir)
        elif "valid_q" in ids:
            # binary cache: id triples + the queries / answers / filter sets of valid and test as CSR lists (what the GPU ranker takes):
            # none of the per-triple Python loops of the text path runs; the (h, r) -> tails dict is rebuilt only if somebody asks
            binary = ids
            self.n_ent, self.n_rel = int(ids["n_ent"]), int(ids["n_rel"])
            as3 = lambda a: np.asarray(a, dtype=np.int64).reshape(-1, 3)
            self.fact_triple, self.train_triple, self.valid_triple, self.test_triple = (as3(ids[k]) for k in ("facts", "train", "valid", "test"))
            self._filters = None
        else:
            self.n_ent, self.n_rel = int(ids["n_ent"]), int(ids["n_rel"])
            self.fact_triple = self._register(ids["facts"])
            self.train_triple = self._register(ids["train"])
            self.valid_triple = self._register(ids["valid"])
            self.test_triple = self._register(ids["test"])

        # add inverse (load_data.py:37-41)
        self.fact_data = self.double_triple(self.fact_triple)
        self.train_data = self.double_triple(self.train_triple)
        self.valid_data = self.double_triple(self.valid_triple)
        self.test_data = self.double_triple(self.test_triple)

        self._graph = self._tgraph = None
        self.load_graph(self.fact_triple)                                               # :43
        self.load_test_graph(np.concatenate([self.fact_triple, self.train_triple], 0))  # :44

        if binary is not None:
            self._csr_host = {}
            for split in ("valid", "test"):
                q, ap, ai = binary[split + "_q"], binary[split + "_ans_ptr"], binary[split + "_ans_idx"]
                setattr(self, split + "_q", [(int(h), int(r)) for h, r in q])
                setattr(self, split + "_a", [ai[ap[i]:ap[i + 1]] for i in range(len(q))])
                self._csr_host[split] = (binary[split + "_filt_ptr"], binary[split + "_filt_idx"])
        else:
            self.valid_q, self.valid_a = self.load_query(self.valid_data)
            self.test_q, self.test_a = self.load_query(self.test_data)
            self._filters = {k: sorted(v) for k, v in self._filters.items()}
        self.n_train, self.n_valid, self.n_test = len(self.train_data), len(self.valid_q), len(self.test_q)
        self._frontiers = {}
        if verbose:
            print("n_train:", self.n_train, "n_valid:", self.n_valid, "n_test:", self.n_test)

    _FILES = ("entities.txt", "relations.txt", "facts.txt", "train.txt", "valid.txt", "test.txt")

    @property
    def filters(self):
        """(h, r) -> sorted tails over all four splits (load_data.py:65-66).  Loaded from the binary cache it is rebuilt on first use."""
        if self._filters is None:
            self._filters = defaultdict(set)
            for t in (self.fact_triple, self.train_triple, self.valid_triple, self.test_triple):
                self._register(t)
            self._filters = {k: sorted(v) for k, v in self._filters.items()}
        return self._filters

    @filters.setter
    def filters(self, value):
        self._filters = value

    def _cached_ids(self, task_dir, cache_dir):
        """Parse the text files once; later runs load `<cache_dir>/<name>_ids.npz` while it is newer than every text file.
        The file holds the id triples and, for valid and test, the queries with their answers and filter sets as CSR lists."""
        os.makedirs(cache_dir, exist_ok=True)
        path = os.path.join(cache_dir, os.path.basename(os.path.normpath(task_dir)) + "_ids.npz")
        newest = max(os.path.getmtime(os.path.join(task_dir, f)) for f in self._FILES)
        if os.path.exists(path) and os.path.getmtime(path) >= newest:
            return dict(np.load(path))
        self._read_text(task_dir)
        ids = dict(n_ent=np.int64(self.n_ent), n_rel=np.int64(self.n_rel), facts=self.fact_triple, train=self.train_triple,
                   valid=self.valid_triple, test=self.test_triple)
        filters = {k: sorted(v) for k, v in self._filters.items()}
        ptr = lambda lists: np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
        cat = lambda lists: np.concatenate([np.asarray(x, dtype=np.int64) for x in lists]) if len(lists) else np.zeros(0, np.int64)
        for split, trip in (("valid", self.valid_triple), ("test", self.test_triple)):
            q, a = self.load_query(self.double_triple(trip))
            fil = [filters[k] for k in q]
            ids.update({split + "_q": np.asarray(q, dtype=np.int64).reshape(-1, 2), split + "_ans_ptr": ptr(a), split + "_ans_idx": cat(a),
                        split + "_filt_ptr": ptr(fil), split + "_filt_idx": cat(fil)})
        tmp = "%s.%d.tmp.npz" % (path, os.getpid())           # ranks may parse concurrently: write aside, then rename into place
        np.savez_compressed(tmp, **ids)
        os.replace(tmp, path)
        self._filters = defaultdict(set)                    # (the text parse above filled it; the binary path starts clean)
        return ids

    # ---- parsing (load_data.py:11-25, 58-67) ----------------------------------------------------
    def _read_text(self, task_dir):
        def ids_of(fname):
            with open(os.path.join(task_dir, fname)) as f:
                return {line.strip(): i for i, line in enumerate(f)}
        self.entity2id = ids_of("entities.txt")
        self.relation2id = ids_of("relations.txt")
        self.n_ent, self.n_rel = len(self.entity2id), len(self.relation2id)
        self.fact_triple = self.read_triples("facts.txt")
        self.train_triple = self.read_triples("train.txt")
        self.valid_triple = self.read_triples("valid.txt")
        self.test_triple = self.read_triples("test.txt")

    def read_triples(self, filename):
        rows = []
        with open(os.path.join(self.task_dir, filename)) as f:
            for line in f:
                h, r, t = line.strip().split()
                rows.append((self.entity2id[h], self.relation2id[r], self.entity2id[t]))
        return self._register(np.array(rows, dtype=np.int64).reshape(-1, 3))

    def _register(self, triples):
        """Every known (h,r,t) of any split goes into the filter sets (load_data.py:65-66)."""
        triples = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
        for h, r, t in triples.tolist():
            self._filters[(h, r)].add(t)
            self._filters[(t, r + self.n_rel)].add(h)
        return triples

    def double_triple(self, triples):
        """load_data.py:69-74."""
        triples = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
        inv = np.stack([triples[:, 2], triples[:, 1] + self.n_rel, triples[:, 0]], 1)
        return np.concatenate([triples, inv], 0)

    # ---- graphs (load_data.py:76-89): inverse + identity rows are added by rg_graph_create -----------
    # The device graphs are built on first use, so the host-side logic of this class (parsing, filters,
    # queries, batches) also runs where there is no GPU.
    def load_graph(self, base_triples):
        if self._graph is not None:
            self._graph.close()
        self.__dict__.pop("_graph_perm", None)       # a permutation left by shuffle_train belongs to ITS triples, not to these
        self._graph, self._graph_base = None, np.asarray(base_triples, dtype=np.int64).reshape(-1, 3)
        self.n_fact = 2 * len(self._graph_base) + self.n_ent

    def load_test_graph(self, base_triples):
        if self._tgraph is not None:
            self._tgraph.close()
        self._tgraph, self._tgraph_base = None, np.asarray(base_triples, dtype=np.int64).reshape(-1, 3)
        self.tn_fact = 2 * len(self._tgraph_base) + self.n_ent

    @property
    def graph(self):
        if self._graph is None:
            perm = self.__dict__.pop("_graph_perm", None)
            if perm is not None and torch.device(self.device).type == "cuda" and torch.cuda.is_available():
                # shuffle_train's re-split on the device: the triples of facts + train live there (uploaded once); an epoch moves
                # only the permutation, and rg_graph_create_device builds the CSRs where they are used
                all_dev = self.__dict__.get("_all_triple_dev")
                if all_dev is None:
                    all_dev = self._all_triple_dev = torch.as_tensor(self._all_triple, dtype=torch.int32).to(self.device)
                idx = torch.as_tensor(perm, dtype=torch.int64).to(self.device)
                self._graph = Graph.from_device(self.n_ent, self.n_rel, all_dev.index_select(0, idx))
            else:
                self._graph = Graph(self.n_ent, self.n_rel, self._graph_base, add_inverse=True, device=self.device)
        return self._graph

    @property
    def tgraph(self):
        if self._tgraph is None:
            self._tgraph = Graph(self.n_ent, self.n_rel, self._tgraph_base, add_inverse=True, device=self.device)
        return self._tgraph

    def graph_for(self, mode):
        """load_data.py:107-112: 'train' walks the fact graph, anything else facts + train."""
        return self.graph if mode == "train" else self.tgraph

    def load_query(self, triples):
        """load_data.py:91-104: group by (h, r), queries sorted by (h, r)."""
        by_hr = defaultdict(list)
        for h, r, t in sorted(map(tuple, np.asarray(triples).tolist()), key=lambda x: (x[0], x[1])):
            by_hr[(h, r)].append(t)
        queries = list(by_hr.keys())
        answers = [np.array(by_hr[k]) for k in queries]
        return queries, answers

    # ---- frontier expansion (load_data.py:106-131) ---------------------------------------------------
    def get_neighbors(self, nodes, mode="train"):
        """Same contract as the reference: nodes [N,2] (batch_idx, entity) ->
        (tail_nodes LongTensor [N',2] sorted, sampled_edges LongTensor [E,6], old_nodes_new_idx LongTensor [N]),
        on the device.  Edges come destination-segmented (the reference's order is fact-row major)."""
        _require_gpu(self.device)
        nodes_t = torch.as_tensor(np.asarray(nodes) if not torch.is_tensor(nodes) else nodes)
        nodes_t = nodes_t.to(device=self.device, dtype=torch.int32).contiguous()
        n_batch = int(nodes_t[:, 0].max().item()) + 1 if nodes_t.numel() else 1
        graph = self.graph_for(mode)
        key = ("gn", n_batch)
        fr = self._frontiers.get(key)
        if fr is None:
            if len(self._frontiers) > 8:           # a handful of batch sizes is the norm; do not hoard workspaces
                self._frontiers.clear()
            fr = self._frontiers[key] = Frontier(self.n_ent, n_batch, 2, self.device)
        fr.reset_nodes(nodes_t)
        fr.expand(graph)
        tail_nodes, _, old_new = fr.nodes(want_prev=False)
        edges, _ = fr.edges(graph, tail_nodes)
        return tail_nodes.long(), edges.long(), old_new.long()

    # ---- batches (load_data.py:133-150) --------------------------------------------------------------
    def get_batch(self, batch_idx, steps=2, data="train"):
        if data == "train":
            return self.train_data[batch_idx]
        query, answer = (self.valid_q, self.valid_a) if data == "valid" else (self.test_q, self.test_a)
        batch_idx = np.asarray(batch_idx)
        subs = np.array([query[i][0] for i in batch_idx])
        rels = np.array([query[i][1] for i in batch_idx])
        objs = np.zeros((len(batch_idx), self.n_ent))
        for i, q in enumerate(batch_idx):
            objs[i][answer[q]] = 1
        return subs, rels, objs

    def _split_csr(self, data):
        """Answers and filter sets of a whole split as device CSR lists, built once."""
        cache = self.__dict__.setdefault("_csr_cache", {})
        if data not in cache:
            query, answer = (self.valid_q, self.valid_a) if data == "valid" else (self.test_q, self.test_a)
            ans = [np.sort(np.asarray(a)) for a in answer]                # np.nonzero order of utils.py:12-13
            ptr = lambda lists: np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
            cat = lambda lists: np.concatenate(lists) if len(lists) else np.zeros(0, np.int64)
            host = self.__dict__.get("_csr_host", {}).get(data)
            if host is not None:                                          # the binary cache holds the filter lists as CSR already
                fptr_h, fidx_h = host
                fil = [fidx_h[fptr_h[i]:fptr_h[i + 1]] for i in range(len(query))]
            else:
                fil = [np.asarray(self.filters[(int(s), int(r))]) for s, r in query]
            to_dev = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(self.device)
            subs, rels = np.array([q[0] for q in query]), np.array([q[1] for q in query])
            cache[data] = (subs, rels, ptr(ans), to_dev(cat(ans), torch.int32), ptr(fil), to_dev(cat(fil), torch.int32),
                           # device copies of the per-query arrays: a contiguous batch is then sliced without any host-to-device copy
                           to_dev(subs, torch.int32), to_dev(rels, torch.int64), to_dev(ptr(ans), torch.int32), to_dev(ptr(fil), torch.int32))
        return cache[data]

    def get_batch_csr(self, batch_idx, data="valid", device_queries=False):
        """The same batch as device CSR lists for the GPU ranker: (subs, rels, ans_ptr, ans_idx, filt_ptr, filt_idx).
        batch_idx must be a contiguous range (as the evaluator's batches are) or any index array.  ``device_queries``: return
        subs / rels of a contiguous batch as device tensors too (the evaluator's loop then issues no host-to-device copy)."""
        subs_all, rels_all, aptr, aidx, fptr, fidx, subs_d, rels_d, aptr_d, fptr_d = self._split_csr(data)
        batch_idx = np.asarray(batch_idx)
        to_dev = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(self.device)
        if len(batch_idx) and np.array_equal(batch_idx, np.arange(batch_idx[0], batch_idx[0] + len(batch_idx))):
            lo, hi = int(batch_idx[0]), int(batch_idx[-1]) + 1
            qs, qr = (subs_d[lo:hi], rels_d[lo:hi]) if device_queries else (subs_all[lo:hi], rels_all[lo:hi])
            return (qs, qr, aptr_d[lo:hi + 1] - int(aptr[lo]), aidx[aptr[lo]:aptr[hi]],
                    fptr_d[lo:hi + 1] - int(fptr[lo]), fidx[fptr[lo]:fptr[hi]])
        seg = lambda ptr, idx: torch.cat([idx[ptr[i]:ptr[i + 1]] for i in batch_idx]) if len(batch_idx) else idx[:0]
        lens = lambda ptr: np.concatenate([[0], np.cumsum([ptr[i + 1] - ptr[i] for i in batch_idx])])
        return subs_all[batch_idx], rels_all[batch_idx], to_dev(lens(aptr)), seg(aptr, aidx), to_dev(lens(fptr)), seg(fptr, fidx)

    def shuffle_train(self):
        """load_data.py:152-164: re-split facts/train 3:1 and rebuild the training graph."""
        if "_all_triple" not in self.__dict__:       # the reference permutes the ORIGINAL facts + train concatenation every epoch
            self._all_triple = np.concatenate([self.fact_triple, self.train_triple], axis=0)
        n_all = len(self._all_triple)
        order = np.random.permutation(n_all)
        all_triple = self._all_triple[order]
        n_f = n_all * 3 // 4
        facts, train = all_triple[:n_f], all_triple[n_f:]
        self.fact_data = self.double_triple(facts)
        self.train_data = self.double_triple(train)
        self.n_train = len(self.train_data)
        self.load_graph(facts)
        self._graph_perm = order[:n_f]               # the training graph is built on the device from the resident triples (see .graph)
