"""HE Wrapper/CryptoTracker.cs: the noise-budget watermark.  The reference compiles the probe in only for DEBUG builds; here it is
off until `CryptoTracker.EnableBudgetTests()` (or CN_BUDGET_TESTS=1), because every probe is a secret-key operation on the client
context plus a host-side CRT of the N coefficients.  A client that can measure budgets offers
`noise_budget(ct_handle, first, count) -> [bits, ...]` (SEAL's integer `Decryptor.InvariantNoiseBudget`; `DeviceClient` does, through
`cn_noise_poly`) or, for a host-side client, `noise_budget_words(ciphertext words) -> bits`."""
import os

import numpy as np

INT_MAX = 2 ** 31 - 1


class CryptoTracker:
    PerformBudgetTests = os.environ.get("CN_BUDGET_TESTS", "0") != "0"
    MinBudgetSoFar = INT_MAX

    @classmethod
    def EnableBudgetTests(cls):
        cls.PerformBudgetTests = True

    @classmethod
    def DisableBudgetTests(cls):
        cls.PerformBudgetTests = False

    @classmethod
    def Reset(cls):
        cls.MinBudgetSoFar = INT_MAX

    @classmethod
    def _note(cls, budget):
        if budget < cls.MinBudgetSoFar:                              # CryptoTracker.cs:45-51
            cls.MinBudgetSoFar = budget
            print("Warning: Current minimal budget %d" % budget)
            if budget == 0:
                raise Exception("error budget is zero")

    @classmethod
    def TestVectorBudget(cls, v, lenv):
        """every ciphertext of every plaintext-prime channel of `v` (:76-89); returns MinBudgetSoFar"""
        for atom, e in zip(v.eVectors, lenv.Environments):
            d = atom.encData
            if d is None:
                continue
            if hasattr(e.client, "noise_budget"):                    # device client: probe where the ciphertext lives
                budgets = e.client.noise_budget(d.h, d.first, d.count)
            elif hasattr(e.client, "noise_budget_words"):            # a host-side client (SEAL in the reference's deployment)
                budgets = [e.client.noise_budget_words(w) for w in e.ctx.ct_download(d.h, d.first, d.count)]
            else:
                continue
            for b in budgets:
                cls._note(int(b))
        return cls.MinBudgetSoFar

    @classmethod
    def TestBudget(cls, res, factory):
        """:58-68: a no-op unless budget tests are on and `res` is an encrypted vector"""
        if not cls.PerformBudgetTests or not hasattr(res, "eVectors"):
            return
        env = factory.AllocateComputationEnv()
        try:
            cls.TestVectorBudget(res, env)
        finally:
            factory.FreeComputationEnv(env)

    @staticmethod
    def Show(x, factory, name="", showAll=False):
        """:98-129: decrypt and print the head of a vector / the corner of a matrix"""
        env = factory.AllocateComputationEnv()
        try:
            dec = np.asarray(x.Decrypt(env))
        finally:
            factory.FreeComputationEnv(env)
        if dec.ndim == 2:
            print("Matrix %s size %dx%d format %s max %.4f" % (name, dec.shape[1], dec.shape[0], x.Format.name, float(np.max(np.abs(dec)))))
            for i in range(min(3, dec.shape[0])):
                print("\t".join("%.4f" % dec[i, j] for j in range(min(3, dec.shape[1]))))
        else:
            last = dec.size if showAll else min(3, dec.size)
            print("%s size %d\t%s\t||\t%.4f\t%.4f" % (name, dec.size, "\t".join("%.4f" % v for v in dec[:last]), float(dec.min()), float(dec.max())))


class OperationsCount:
    """`OperationsCount` (HE Wrapper/AtomicSealBfvVector.cs:208-294): how many evaluator operations of each kind ran.  The counters
    live in libcnhip, one set per context (`cn_stats_get`, same names; plus the limb transforms and kernel launches behind them);
    this class sums them over the plaintext primes of a factory, per prime like the reference's per-evaluator-call counting."""
    Totals = {}

    @staticmethod
    def _contexts(factory):
        return [e.ctx for e in getattr(factory.AllocateComputationEnv(), "Environments", ()) if hasattr(e.ctx, "stats")]

    @classmethod
    def Counts(cls, factory, reset=False):
        out = {}
        for ctx in cls._contexts(factory):
            for k, v in ctx.stats(reset=reset).items():
                out[k] = out.get(k, 0) + v
        return out

    @classmethod
    def Reset(cls, factory):
        for k, v in cls.Counts(factory, reset=True).items():          # :254-270: fold into the totals, then zero
            cls.Totals[k] = cls.Totals.get(k, 0) + v

    @classmethod
    def Print(cls, factory):
        print("Operations:")
        for k, v in cls.Counts(factory).items():
            print("\t%s\t%d" % (k, v))

    @classmethod
    def PrintTotals(cls, factory):
        print("Operations (total):")
        for k, v in cls.Counts(factory).items():
            print("\t%s\t%d" % (k, v + cls.Totals.get(k, 0)))
