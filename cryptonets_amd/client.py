"""Client-side crypto on the device: KeyGenerator / Encryptor / Decryptor through libcnhip (cn_keygen / cn_encrypt / cn_decrypt).

This is the data owner's side (it holds the secret key); an evaluation server only ever receives the public evaluation keys and
ciphertexts.  With this client the package is self-contained: no SEAL and no CPU oracle is needed to run a network end to end.
"""
import numpy as np

from .hewrapper import ClientCrypto


class DeviceClient(ClientCrypto):
    def __init__(self, ctx, seed=0x5EA1):
        self.ctx, self.seed = ctx, seed

    def generate_keys(self, with_galois=True):
        self.ctx.keygen(self.seed, galois=with_galois)          # also installs the evaluation keys in this context
        self.with_galois = with_galois

    def relin_key(self):
        return None                                              # already resident in HBM (see AtomicSealBfvEncryptedEnvironment)

    def galois_keys(self):
        return {}

    # device-resident fast paths used by the wrapper
    def encrypt_device(self, pt_handle, first, count, out_handle, out_first):
        self.ctx.encrypt(pt_handle, first, out_handle, out_first, count, seed=self.seed)

    def decrypt_device(self, ct_handle, first, count):
        pt = self.ctx.pt_alloc(count)
        try:
            self.ctx.decrypt(ct_handle, first, count, pt, 0)
            return self.ctx.pt_download(pt, 0, count)
        finally:
            self.ctx.free(pt)

    def noise_budget(self, ct_handle, first=0, count=1):
        """Decryptor.InvariantNoiseBudget of `count` ciphertexts (integer bits, like SEAL): what CryptoTracker probes"""
        return self.ctx.invariant_noise_budget(ct_handle, first, count, exact_bits=True)

    # ClientCrypto interface on host arrays
    def encrypt(self, plain):
        pt, ct = self.ctx.pt_alloc(1), self.ctx.ct_alloc(1)
        try:
            self.ctx.pt_upload(pt, 0, np.ascontiguousarray(plain, dtype=np.uint64))
            self.ctx.encrypt(pt, 0, ct, 0, 1, seed=self.seed)
            return self.ctx.ct_download(ct, 0, 1)[0]
        finally:
            self.ctx.free(pt)
            self.ctx.free(ct)

    def decrypt(self, ctwords):
        w = np.ascontiguousarray(ctwords, dtype=np.uint64)
        size = w.size // (self.ctx.k * self.ctx.n)
        ct = self.ctx.ct_alloc(1, size)
        try:
            self.ctx.ct_upload(ct, 0, w[None, :])
            return self.decrypt_device(ct, 0, 1)[0]
        finally:
            self.ctx.free(ct)
