"""Generates tests/golden/cryptonets_weights.npz from the reference's model weights (data, not code):
`CryptoNets/Weights.cs` (Weights_0[130] :8, Weights_1[84500] :24, Weights_3[1000] :873, Biases_2[100] :976, Biases_3[10] :978).
Run in the build container only (reads /root/reference); the GPU box uses the committed .npz."""
import re
import numpy as np

src = open("/root/reference/CryptoNets/Weights.cs").read()
src = re.sub(r"//[^\n]*", "", src)
out = {}
for name in ("Weights_0", "Weights_1", "Weights_3", "Biases_2", "Biases_3"):
    m = re.search(name + r"\s*\{\s*get;\s*\}\s*=\s*new\s+double\[\]\s*\{(.*?)\};", src, re.S)
    out[name] = np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()], dtype=np.float64)
    print(name, out[name].shape)
assert out["Weights_0"].size == 130 and out["Weights_1"].size == 84500 and out["Weights_3"].size == 1000
assert out["Biases_2"].size == 100 and out["Biases_3"].size == 10
np.savez_compressed("/root/repo/tests/golden/cryptonets_weights.npz", **out)

# SmallLoLa's model (`LowLatencyCryptoNets/SmallModel.cs`: Weights_0[130], Weights_1[8450], Biases_1[10]; "accuracy: 0.96943")
src = re.sub(r"//[^\n]*", "", open("/root/reference/LowLatencyCryptoNets/SmallModel.cs").read())
small = {}
for name in ("Weights_0", "Weights_1", "Biases_1"):
    m = re.search(name + r"\s*\{\s*get;\s*\}\s*=\s*new\s+double\[\]\s*\{(.*?)\};", src, re.S)
    small[name] = np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()], dtype=np.float64)
    print("SmallModel", name, small[name].shape)
assert small["Weights_0"].size == 130 and small["Weights_1"].size == 8450 and small["Biases_1"].size == 10
np.savez_compressed("/root/repo/tests/golden/small_model_weights.npz", **small)
