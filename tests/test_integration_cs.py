"""The C# side of the boundary (integration/*.cs) cannot be compiled in this image (no .NET toolchain).  What CAN be checked is checked:
the P/Invoke file is regenerated from include/cnhip.h and must equal the committed one; every libcnhip call the twin makes exists in
the header with that many arguments; every member the UNCHANGED reference files use on the atomic classes is defined by the twin."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
INTEG = os.path.join(ROOT, "integration")
TWIN = os.path.join(INTEG, "GpuAtomicSealBfvEncryptedVector.cs")
FACTORY = os.path.join(INTEG, "GpuSealBfvFactory.cs")

# members of AtomicSealBfvEncryptedEnvironment / AtomicSealBfvEncryptedVector / OperationsCount that EncryptedSealBfvVector.cs,
# EncryptedSealBfvMatrix.cs, IFactory.cs, CryptoTracker.cs, BaseLayer.cs and CryptoNets.cs use (collected from /root/reference)
USED_MEMBERS = ["Dim", "plainmodulusValue", "Write", "Scale", "GenerateEncryptionKeys", "BlockSize", "decryptor", "SumAllSlotsTask", "SumAllSlots",
                "SubtractTask", "StackTask", "SaveToStream", "RotateTask", "Read", "PointwiseMultiplyTask", "PermuteTask", "ParentFactory",
                "LoadFromStream", "IsEncrypted", "InterleaveTask", "GenerateSparseOfArray", "Format", "DuplicateTask", "DotProductTask",
                "DenseMatrixBySparseVectorMultiplyTask", "Decrypt", "AddTask", "DecryptFullPrecisionTask", "DecryptTask", "SparseMultiplyTask",
                "RegisterDim", "RegisterScale", "Data", "IsSigned", "Dispose", "parameters", "Primes", "AllocateComputationEnv", "FreeComputationEnv",
                "Reset", "Print", "PrintTotals"]
# every method of the IVector interface (HE Wrapper/IVector.cs:14-40)
IVECTOR = ["Decrypt", "DecryptFullPrecision", "Write", "Data", "Subtract", "Add", "DotProduct", "PointwiseMultiply", "SumAllSlots", "Duplicate", "Rotate",
           "Permute", "Dim", "Scale", "RegisterScale", "IsEncrypted", "IsSigned", "BlockSize", "Format"]


def _calls(src):
    """[(name, number of arguments)] of every CnHip.cn_*( ... ) call in a C# source"""
    out = []
    for m in re.finditer(r"CnHip\.(cn_[a-z0-9_]+)\(", src):
        i, depth, args, cur = m.end(), 1, 0, ""
        while depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            elif ch == "," and depth == 1:
                args += 1
                cur = ""
                i += 1
                continue
            cur += ch
            i += 1
        out.append((m.group(1), args + (1 if cur.strip() or args else 0)))
    return out


def test_pinvoke_file_is_generated_from_the_header():
    import gen_cs_pinvoke
    assert open(gen_cs_pinvoke.TARGET).read() == gen_cs_pinvoke.generate(), "run python tools/gen_cs_pinvoke.py"
    names = [n for _, n, _ in gen_cs_pinvoke.prototypes()]
    from cryptonets_amd import _native
    assert sorted(names) == sorted(_native.SIGNATURES)


@pytest.mark.parametrize("path", [TWIN, FACTORY])
def test_twin_calls_match_the_c_abi(path):
    import gen_cs_pinvoke
    protos = {n: len(p) for _, n, p in gen_cs_pinvoke.prototypes()}
    src = open(path).read()
    calls = _calls(src)
    assert calls, "no libcnhip calls found"
    for name, nargs in calls:
        assert name in protos, "%s is not declared in include/cnhip.h" % name
        assert nargs == protos[name], "%s called with %d arguments, the C ABI takes %d" % (name, nargs, protos[name])


def test_twin_covers_the_hot_path_entry_points():
    """the evaluator calls of SURVEY 2a / 8a must each be reachable from the twin"""
    used = {n for n, _ in _calls(open(TWIN).read())}
    for need in ("cn_scalar_dot", "cn_mul_relin", "cn_add", "cn_sub", "cn_add_plain", "cn_mul_plain", "cn_mul_scalar", "cn_add_many", "cn_rotate_rows",
                 "cn_rotate_rows_add", "cn_rotate_columns", "cn_rotate_columns_add", "cn_sum_slots", "cn_copy", "cn_ct_upload", "cn_ct_download",
                 "cn_pt_upload", "cn_encode", "cn_set_relin_key", "cn_set_galois_key", "cn_ctx_create", "cn_ctx_destroy", "cn_free", "cn_ct_alloc", "cn_pt_alloc"):
        assert need in used, need
    assert "cn_ctx_broadcast_keys" in {n for n, _ in _calls(open(FACTORY).read())}


def test_twin_runs_the_start_up_self_test():
    """CreateDevice ends with SelfTest(): SEAL's own Evaluator against the device, both key-switch conventions, coefficient-form fallback,
    an exception when nothing matches (the Python mirror of the same procedure is executed by tests/test_self_test.py)"""
    src = open(TWIN).read()
    body = src[src.index("void CreateDevice("):src.index("public string SelfTestReport")]
    assert body.rstrip().endswith("SelfTest();\n        }".replace("\n", "\n")) or "SelfTest();" in body
    st = src[src.index("public void SelfTest()"):src.index("void UploadKeysInCoefficientForm()")]
    for op in ("MultiplyPlain", "MultiplyPlain(constant)", "AddPlain", "Multiply", "Relinearize", "RotateRows(1)", "RotateRows(-1)", "RotateColumns"):
        assert st.count('"%s"' % op) >= 3, op                      # device side, SEAL side, the list that is walked
    assert 'cn_set_option(device.Ctx, "ks_xi"' in st and "UploadKeysInCoefficientForm()" in st and st.count("throw new Exception") == 2
    up = src[src.index("void UploadKeysInCoefficientForm()"):]
    assert up.count("cn_load_key(") >= 3 and "TransformFromNTTInplace" in up
    # the first device nonce is not derived from the sampler key
    assert "BitConverter.ToInt64(rngKey" not in src


def test_twin_defines_every_member_the_unchanged_files_use():
    src = open(TWIN).read()
    for name in USED_MEMBERS + IVECTOR:
        assert re.search(r"\b%s\b\s*(\(|\{|=|;|,)" % re.escape(name), src), "member %s is missing from the twin" % name
    for cls in ("class AtomicSealBfvEncryptedEnvironment : IComputationEnvironment", "class AtomicSealBfvEncryptedVector : IVector", "static class OperationsCount"):
        assert cls in src
    assert src.count("{") == src.count("}") and src.count("(") == src.count(")")
    f = open(FACTORY).read()
    assert "class GpuSealBfvFactory : EncryptedSealBfvFactory" in f and f.count("{") == f.count("}")


@pytest.mark.skipif(not os.path.isdir("/root/reference/HE Wrapper"), reason="the reference checkout is only present in the build container")
def test_member_list_is_current_with_the_reference():
    """every `x.Member` the unchanged wrapper files apply to the atomic classes is in USED_MEMBERS"""
    base = "/root/reference/HE Wrapper"
    found = set()
    for f in ("EncryptedSealBfvVector.cs", "EncryptedSealBfvMatrix.cs", "IFactory.cs", "CryptoTracker.cs"):
        txt = open(os.path.join(base, f), encoding="utf-8-sig").read()
        found |= set(re.findall(r"eVectors\[[a-z0-9]*\]\.([A-Za-z]+)", txt))
        found |= set(re.findall(r"AtomicSealBfvEncryptedVector\.([A-Za-z]+)", txt))
        found |= set(re.findall(r"Environments\[[a-z0-9]*\]\.([A-Za-z]+)", txt))
        found |= set(re.findall(r"envs\[i\]\.([A-Za-z]+)", txt))
    assert found <= set(USED_MEMBERS), found - set(USED_MEMBERS)


# ---- the SEALNet 3.2 surface of the twin: integration/SEALNET_MANIFEST.json
SEAL_TYPES = ["EncryptionParameters", "SchemeType", "SmallModulus", "DefaultParams", "SEALContext", "MemoryPoolHandle", "KeyGenerator", "Evaluator", "Encryptor",
              "Decryptor", "BatchEncoder", "Plaintext", "Ciphertext", "PublicKey", "SecretKey", "RelinKeys", "GaloisKeys", "BigUInt", "ParmsId", "IntegerEncoder",
              "CKKSEncoder", "KSwitchKeys", "Serialization"]
RECOLLECTED = {("SEALContext", "FirstParmsId"), ("Ciphertext", "UInt64Count"), ("Ciphertext", "this[ulong] get/set"), ("Ciphertext", "Resize(context, parmsId, size)"),
               ("Ciphertext", "IsNTTForm"), ("Plaintext", "this[ulong]"), ("PublicKey", "Data"), ("RelinKeys", "Data"), ("RelinKeys", "DecompositionBitCount"),
               ("GaloisKeys", "Data"), ("GaloisKeys", "DecompositionBitCount"), ("Evaluator", "TransformFromNTTInplace(encryptedNTT)")}


def _manifest():
    import json
    return json.load(open(os.path.join(INTEG, "SEALNET_MANIFEST.json")))["members"]


def test_sealnet_manifest_covers_the_twin():
    """every SEAL type the twin names has its members in the manifest, every manifest entry is really used by the twin, and the members
    that are NOT backed by a use in the reference's own sources are exactly the known eleven (a new unbacked member cannot slip in)"""
    src = open(TWIN).read() + open(FACTORY).read()
    code = re.sub(r"//[^\n]*", "", src)
    members = _manifest()
    types = {m["type"] for m in members}
    for t in SEAL_TYPES:
        if re.search(r"\b%s\b" % t, code):
            assert t in types, "SEAL type %s is used by the twin but has no manifest entry" % t
    for m in members:
        assert re.search(m["twin_pattern"], code), "manifest entry %s.%s is not used by the twin any more" % (m["type"], m["member"])
        assert m["seal_3_2_source"].startswith("dotnet/src/")
    unbacked = {(m["type"], m["member"]) for m in members if "reference_use" not in m}
    assert unbacked == RECOLLECTED, unbacked ^ RECOLLECTED
    # the stream fast path of round 2 (an assumed 73-byte Ciphertext.Save header) is gone: words travel through the public indexer only
    assert "CtHeader" not in src and "mem.GetBuffer()" not in src


@pytest.mark.skipif(not os.path.isdir("/root/reference/HE Wrapper"), reason="the reference checkout is only present in the build container")
def test_sealnet_manifest_reference_backing_is_real():
    """`used by the reference itself` entries: the cited line of the reference's own source shows the member"""
    for m in _manifest():
        if "reference_use" not in m:
            continue
        f, line = m["reference_use"].rsplit(":", 1)
        lines = open(os.path.join("/root/reference", f), encoding="utf-8-sig").read().splitlines()
        assert re.search(m["reference_pattern"], lines[int(line) - 1]), (m["type"], m["member"], lines[int(line) - 1])


@pytest.mark.parametrize("path", ["GpuAtomicSealBfvEncryptedVector.cs", "GpuSealBfvFactory.cs", "CnHip.cs"])
def test_twin_sources_are_bracket_balanced(path):
    """no .NET toolchain in the image: the least a compiler would check - braces, parentheses and brackets of the twin's sources balance
    (comments, strings and character literals stripped) and never go negative"""
    src = open(os.path.join(ROOT, "integration", path)).read()
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r'"(\\.|[^"\\])*"', '""', src)
    src = re.sub(r"'(\\.|[^'\\])'", "''", src)
    depth = {"{": 0, "(": 0, "[": 0}
    close = {"}": "{", ")": "(", "]": "["}
    for ch in src:
        if ch in depth:
            depth[ch] += 1
        elif ch in close:
            depth[close[ch]] -= 1
            assert depth[close[ch]] >= 0, "unbalanced %s in %s" % (ch, path)
    assert all(v == 0 for v in depth.values()), depth
