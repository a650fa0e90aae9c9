"""CPU test harness: the libcnhip `Context` protocol and the `ClientCrypto` interface implemented on the oracle.

Lets the SAME wrapper logic (cryptonets_amd/hewrapper.py) replay the reference's known-answer tests on the CPU
(pinning the oracle at the decrypted-slot level) and provides the client side (keygen / encrypt / decrypt) for the
GPU tests.  Lives under tests/: the product never imports it.
"""
import numpy as np

from oracle.cno import COEFF_MODULUS_128, Oracle


class OracleClient:
    """ClientCrypto on the oracle (KeyGenerator / Encryptor / Decryptor of one plaintext modulus)."""

    def __init__(self, t, n, q, dbc, gdbc, seed=None, oracle=None):
        self.o = oracle if oracle is not None else Oracle(n, t, q=q, dbc=dbc, gdbc=gdbc)
        self.seed = (t % 1000003) if seed is None else seed

    def generate_keys(self, with_galois=True):
        self.o.keygen(self.seed, galois=with_galois)

    def relin_key(self):
        return self.o.relin_key()

    def galois_keys(self):
        return {e: self.o.galois_key(i) for i, e in enumerate(self.o.galois_elts())}

    def encrypt(self, plain):
        return self.o.encrypt(np.ascontiguousarray(plain, dtype=np.uint64))

    def decrypt(self, ct):
        return self.o.decrypt(np.ascontiguousarray(ct, dtype=np.uint64))

    # the client's own evaluator and its keys in coefficient form: what AtomicSealBfvEncryptedEnvironment.SelfTest compares the device with
    def reference_evaluator(self):
        return self.o

    def relin_key_coeff_form(self):
        return self.o.key_to_coeff_form(self.o.relin_key())

    def galois_keys_coeff_form(self):
        return {e: self.o.key_to_coeff_form(self.o.galois_key(i)) for i, e in enumerate(self.o.galois_elts())}

    def noise_poly(self, ct):
        """t * (c0 + c1 s + c2 s^2) mod q_j, [k, n]: the polynomial whose centred norm Decryptor.InvariantNoiseBudget measures"""
        o = self.o
        x = o.dot_with_secret(np.ascontiguousarray(ct, dtype=np.uint64)).reshape(o.k, o.n)
        return np.stack([np.array([(int(v) * o.t) % int(o.q[j]) for v in x[j]], dtype=np.uint64) for j in range(o.k)])

    def noise_budget_words(self, ct):
        """SEAL's integer budget: max(0, bitcount(q) - bitcount(norm) - 1)"""
        o = self.o
        w = self.noise_poly(ct)
        Q = 1
        for qj in o.q:
            Q *= int(qj)
        coef = [(Q // int(qj)) * pow((Q // int(qj)) % int(qj), -1, int(qj)) for qj in o.q]
        x = sum(w[j].astype(object) * coef[j] for j in range(o.k)) % Q
        norm = max(int(v) if 2 * int(v) <= Q else Q - int(v) for v in x)
        return max(0, Q.bit_length() - norm.bit_length() - 1)


class OracleBackend:
    """Same method names as cryptonets_amd._native.Context, computing with the oracle on numpy arrays."""

    def __init__(self, oracle):
        self.o = oracle
        self.n, self.t, self.q, self.k = oracle.n, oracle.t, list(oracle.q), oracle.k
        self.ctw = oracle.ctw
        self.dbc, self.gdbc = oracle.dbc, oracle.gdbc
        self.bufs, self.next = {}, 1
        self._recording, self._graphs, self._parked, self._depth = None, {}, set(), 0

    # keys (cn_get_key / cn_has_galois_key)
    def get_key(self, which, elt=0):
        if which == 0:
            return self.o.relin_key()
        if which == 1:
            return self.o.galois_key(self.o.galois_elts().index(elt))
        return self.o.public_key() if which == 2 else self.o.secret_key()

    def has_galois_key(self, elt):
        return elt in self.o.galois_elts()

    # buffers
    def ct_alloc(self, count, size=2):
        h = self.next
        self.next += 1
        self.bufs[h] = np.zeros((count, size * self.k * self.n), dtype=np.uint64)
        return h

    def pt_alloc(self, count):
        h = self.next
        self.next += 1
        self.bufs[h] = np.zeros((count, self.n), dtype=np.uint64)
        return h

    def free(self, h):
        if h in self._graphs:
            for g in self._graphs.pop(h)[1]:
                self.bufs.pop(g, None)
            return
        if self._recording is not None:                 # arrays freed while recording stay reserved for the graph (cn_graph_end)
            self._recording[1].append(h)
            self._parked.add(h)
            return
        del self.bufs[h]
        self._parked.discard(h)

    def live_handles(self):
        return len(self.bufs) - len(self._parked) + len(self._graphs)

    # ---- emulation of cn_graph_begin / cn_graph_end / cn_graph_launch for the CPU suite: the compute calls made while recording are
    # logged (and executed), a launch re-executes them on the same handles - what the HIP graph does with the same device addresses.
    # Uploads / downloads / synchronisation are refused while recording, like the library does.
    _REPLAYED = ("copy", "copy_many", "rotate_rows_many", "add", "sub", "add_many", "add_plain", "mul_plain", "mul_scalar", "scalar_gemm", "gemm_apply", "mul_relin", "multiply",
                 "relinearize", "rotate_rows", "rotate_rows_add", "rotate_columns", "rotate_columns_add", "sum_slots", "rowdot_batch")
    _REFUSED = ("sync", "ct_upload", "ct_download", "pt_upload", "pt_download", "encode", "decode", "encode_batch", "decode_batch", "set_relin_key", "set_galois_key")

    def graph_begin(self):
        if self._recording is not None:
            raise RuntimeError("cn_graph_begin is not possible while a graph is recorded")
        self._recording = ([], [])
        self.__class__ = _RecordingBackend

    def graph_end(self):
        calls, parked = self._recording
        self._recording = None
        self.__class__ = OracleBackend
        h = self.next
        self.next += 1
        self._graphs[h] = (calls, parked)
        return h

    def graph_launch(self, graph):
        for name, args, kw in self._graphs[graph][0]:
            getattr(OracleBackend, name)(self, *args, **kw)


    def ct_upload(self, h, first, data):
        d = np.asarray(data, dtype=np.uint64).reshape(-1, self.bufs[h].shape[1])
        self.bufs[h][first:first + len(d)] = d

    def ct_download(self, h, first, count, size=2):
        return self.bufs[h][first:first + count].copy()

    def pt_upload(self, h, first, data):
        d = np.asarray(data, dtype=np.uint64).reshape(-1, self.n)
        self.bufs[h][first:first + len(d)] = d

    def pt_download(self, h, first, count):
        return self.bufs[h][first:first + count].copy()

    def encode(self, values, pt, pi):
        self.bufs[pt][pi] = self.o.encode(np.asarray(values, dtype=np.uint64))

    def decode(self, pt, pi):
        return self.o.decode(self.bufs[pt][pi])

    def encode_batch(self, values, pt, pi):
        for c, row in enumerate(np.asarray(values, dtype=np.uint64)):
            self.encode(row, pt, pi + c)

    def decode_batch(self, pt, pi, count):
        return np.stack([self.decode(pt, pi + c) for c in range(count)])

    def copy(self, src, sfirst, dst, dfirst, count):
        self.bufs[dst][dfirst:dfirst + count] = self.bufs[src][sfirst:sfirst + count].copy()

    def copy_many(self, srcs, sfirsts, dst, dfirst):
        for i, (h, f) in enumerate(zip(srcs, sfirsts)):
            self.bufs[dst][dfirst + i] = self.bufs[int(h)][int(f)].copy()

    def set_relin_key(self, words):
        pass            # the oracle context already holds the keys it generated

    def set_galois_key(self, elt, words):
        pass

    def load_key(self, which, words, elt=0, coeff_form=False):
        pass

    def get_option(self, name):
        if name == "ks_xi":
            return int(self.o.ks_xi)
        raise KeyError(name)

    def set_option(self, name, value):
        if name != "ks_xi":
            raise KeyError(name)
        self.o.set_ks_xi(bool(value))

    def multiply(self, a, ai, b, bi, out3, oi, count=1):
        for i in range(count):
            self.bufs[out3][oi + i] = self.o.multiply(self.bufs[a][ai + i], self.bufs[b][bi + i])

    def relinearize(self, in3, ii, out, oi, count=1):
        for i in range(count):
            self.bufs[out][oi + i] = self.o.relinearize(self.bufs[in3][ii + i])

    def sync(self):
        pass

    # evaluator
    def add(self, a, ai, b, bi, out, oi, count=1):
        for i in range(count):
            self.bufs[out][oi + i] = self.o.add(self.bufs[a][ai + i], self.bufs[b][bi + i])

    def sub(self, a, ai, b, bi, out, oi, count=1):
        for i in range(count):
            self.bufs[out][oi + i] = self.o.sub(self.bufs[a][ai + i], self.bufs[b][bi + i])

    def add_many(self, src, idx, out, oi):
        acc = self.bufs[src][idx[0]].copy()
        for i in idx[1:]:
            acc = self.o.add(acc, self.bufs[src][i])
        self.bufs[out][oi] = acc

    def add_plain(self, a, ai, pt, pi, out, oi, count=1, subtract=False):
        for i in range(count):
            self.bufs[out][oi + i] = self.o.add_plain(self.bufs[a][ai + i], self.bufs[pt][pi + i], subtract)

    def mul_plain(self, a, ai, pt, pi, out, oi, count=1, pt_stride=1):
        for i in range(count):
            self.bufs[out][oi + i] = self.o.multiply_plain(self.bufs[a][ai + i], self.bufs[pt][pi + i * pt_stride])

    def mul_scalar(self, a, ai, scalars, out, oi, count=1, broadcast=False):
        s = np.asarray(scalars, dtype=np.uint64).reshape(-1)
        for i in range(count):
            self.bufs[out][oi + i] = self.o.multiply_plain(self.bufs[a][ai + i], s[0:1] if broadcast else s[i:i + 1])

    def scalar_gemm(self, src, W, out, oi, idx=None, bias_pt=0, bias_idx=None):
        res = self.o.scalar_gemm(self.bufs[src], W, idx)
        if bias_pt:
            res = self.o.add_plain_batch(res, self.bufs[bias_pt][np.asarray(bias_idx)])
        self.bufs[out][oi:oi + len(res)] = res

    def mul_relin(self, a, ai, b, bi, out, oi, count=1, a_stride=1, b_stride=1):
        for i in range(count):
            m3 = self.o.multiply(self.bufs[a][ai + i * a_stride], self.bufs[b][bi + i * b_stride])
            self.bufs[out][oi + i] = self.o.relinearize(m3)

    def rotate_rows(self, src, ii, steps, out, oi, count=1):
        for i in range(count):
            self.bufs[out][oi + i] = self.o.rotate_rows(self.bufs[src][ii + i], steps)

    def rotate_rows_many(self, src, iis, steps, out, ois):
        res = [self.o.rotate_rows(self.bufs[src][int(i)], int(s_)) if int(s_) else self.bufs[src][int(i)].copy() for i, s_ in zip(iis, steps)]
        for r, o_ in zip(res, ois):
            self.bufs[out][int(o_)] = r

    def rotate_rows_add(self, src, ii, steps, acc, ai, out, oi, count=1):
        for c in range(count):
            self.bufs[out][oi + c] = self.o.add(self.bufs[acc][ai + c], self.o.rotate_rows(self.bufs[src][ii + c], steps))

    def rotate_columns_add(self, src, ii, acc, ai, out, oi, count=1):
        for c in range(count):
            self.bufs[out][oi + c] = self.o.add(self.bufs[acc][ai + c], self.o.rotate_columns(self.bufs[src][ii + c]))

    def sum_slots(self, h, first, count, length=0):
        half = self.n // 2
        ln = length if length else self.n
        if ln >= half:
            self.rotate_columns_add(h, first, h, first, h, first, count)
            ln = half
        steps = 1
        while steps < ln:
            self.rotate_rows_add(h, first, -steps, h, first, h, first, count)
            steps *= 2

    def rowdot_batch(self, v, vi, pt, pi, rows, length, out, oi):
        for r in range(rows):
            self.bufs[out][oi + r] = self.o.multiply_plain(self.bufs[v][vi], self.bufs[pt][pi + r])
        if length != 1:
            self.sum_slots(out, oi, rows, length)

    def rotate_columns(self, src, ii, out, oi, count=1):
        for i in range(count):
            self.bufs[out][oi + i] = self.o.rotate_columns(self.bufs[src][ii + i])


class OracleDeviceBackend(OracleBackend):
    """An OracleBackend with an oracle instance of its OWN that receives the evaluation keys the way a device does (set_relin_key /
    set_galois_key / load_key) and has its own "ks_xi" switch - the device side of the CPU tests of the start-up self-test, where the
    client may follow the other key-switch convention or hand over NTT-form keys in another transform order."""

    def set_relin_key(self, words):
        self.o.import_relin_key(words)

    def set_galois_key(self, elt, words):
        self.o.import_galois_key(elt, words)

    def load_key(self, which, words, elt=0, coeff_form=False):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        if coeff_form:                                  # the device transforms with ITS tables
            w = w.reshape(-1, self.k, self.n).copy()
            for i in range(w.shape[0]):
                for j in range(self.k):
                    w[i, j] = self.o.ntt_fwd(j, w[i, j])
            w = w.reshape(-1)
        if which == 0:
            self.o.import_relin_key(w)
        elif which == 1:
            self.o.import_galois_key(elt, w)
        else:
            raise NotImplementedError


class _RecordingBackend(OracleBackend):
    """OracleBackend while a graph is recorded (graph_begin swaps the instance's class, graph_end swaps it back): compute calls are
    logged and executed, synchronising calls are refused.  Kept off the normal class - attribute interception is slow."""

    def __getattribute__(self, name):
        attr = object.__getattribute__(self, name)
        if name.startswith("_") or not callable(attr):
            return attr
        rec = object.__getattribute__(self, "_recording")
        if rec is None:
            return attr
        if name in OracleBackend._REFUSED:
            def refused(*a, **k):
                raise RuntimeError("%s is not possible while a graph is recorded" % name)
            return refused
        if name in OracleBackend._REPLAYED:
            def logged(*a, **k):
                if object.__getattribute__(self, "_depth"):     # a compound call (rowdot_batch, sum_slots) is logged once, not its parts
                    return attr(*a, **k)
                rec[0].append((name, tuple(np.array(x, copy=True) if isinstance(x, (np.ndarray, list)) else x for x in a),
                               {kk: (np.array(v, copy=True) if isinstance(v, (np.ndarray, list)) else v) for kk, v in k.items()}))
                self._depth = 1
                try:
                    return attr(*a, **k)
                finally:
                    self._depth = 0
            return logged
        return attr


class OracleHarness:
    """Factories for EncryptedSealBfvFactory(client_factory=..., context_factory=...): backend 'gpu' = libcnhip contexts with
    oracle-made keys; backend 'cpu' = everything on the oracle."""

    def __init__(self, backend, ks_xi=False):
        self.backend = backend
        self.ks_xi = ks_xi             # the CLIENT's key-switch convention (the device finds it out in its start-up self-test)
        self.oracles = {}

    def default_coeff_modulus(self, n):
        return list(COEFF_MODULUS_128[n])

    def _oracle(self, n, t, q, dbc, gdbc):
        key = (n, t, tuple(q), dbc, gdbc)
        if key not in self.oracles:
            self.oracles[key] = Oracle(n, t, q=q, dbc=dbc, gdbc=gdbc, ks_xi=self.ks_xi)
        return self.oracles[key]

    def client_factory(self, t, n, q, dbc, gdbc):
        return OracleClient(t, n, q, dbc, gdbc, oracle=self._oracle(n, t, q, dbc, gdbc))

    def __call__(self, n, t, q, dbc, gdbc):          # context_factory
        if self.backend == "cpu":
            return OracleBackend(self._oracle(n, t, q, dbc, gdbc))
        from cryptonets_amd._native import Context
        return Context(n, t, q=q, dbc=dbc, gdbc=gdbc, device=0)


def make_factory(backend, primes=None, n=4096, dbc=10, gdbc=20, small_modulus_count=-1, galois=True, ks_xi=False):
    from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
    h = OracleHarness(backend, ks_xi=ks_xi)
    return EncryptedSealBfvFactory(primes, n, dbc, gdbc, small_modulus_count, client_factory=h.client_factory, context_factory=h, galois=galois)
