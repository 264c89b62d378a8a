"""Transcribe the data of TEST(KeySwitch, small) (test/experimental/seal/test-key-switch.cpp:16-190)
into tests/golden/hexl_kat.json: numeric inputs and the expected output only.
Run in the build container (reads /root/reference); the GPU box uses the committed JSON."""
import json
import os
import re

SRC = "/root/reference/test/experimental/seal/test-key-switch.cpp"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = open(SRC).read()


def block(name):
    """Numbers inside `name{ ... };` (possibly nested one level)."""
    start = text.index(name + "{")
    depth, i = 0, start + len(name)
    while True:
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    return text[start + len(name):i + 1]


def numbers(s):
    return [int(x) for x in re.findall(r"\d+", s)]


keys_src = block("key_vector")
inner = re.findall(r"\{([^{}]*)\}", keys_src)
keys = [numbers(b) for b in inner]
case = {
    "n": numbers(re.search(r"coeff_count = (\d+)", text).group(0))[0],
    "moduli": numbers(block("moduli")),
    "modswitch_factors": numbers(block("modswitch_factors")),
    "decomp_modulus_size": int(re.search(r"decomp_modulus_size = (\d+)", text).group(1)),
    "key_modulus_size": int(re.search(r"key_modulus_size = (\d+)", text).group(1)),
    "rns_modulus_size": int(re.search(r"rns_modulus_size = (\d+)", text).group(1)),
    "key_component_count": int(re.search(r"key_component_count = (\d+)", text).group(1)),
    "keys": keys,
    "input": numbers(block("input")),
    "t_target": numbers(block("t_target_iter_ptr")),
    "out": numbers(block("expected_output")),
}
n = case["n"]
assert len(keys) == case["decomp_modulus_size"]
assert all(len(k) == case["key_component_count"] * case["key_modulus_size"] * n for k in keys)
assert len(case["t_target"]) == case["decomp_modulus_size"] * n
path = os.path.join(ROOT, "tests", "golden", "hexl_kat.json")
d = json.load(open(path))
d["key_switch"] = {
    "source": "test/experimental/seal/test-key-switch.cpp:16-190 TEST(KeySwitch, small); the "
              "test compares the first key_component_count * decomp_modulus_size * n words of "
              "`input` (the accumulated result) -- the tail of `input` is never written",
    "cases": [case]}
json.dump(d, open(path, "w"), indent=1)
print({k: (len(v) if isinstance(v, list) else v) for k, v in case.items()})
