"""GPU debugging helper: device fe_mul vs Python integers on edge inputs; prints mismatch patterns."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonky_amd import api, synth
from oracle import bigint_ref as br
from tests.util import array_to_ints, ints_to_array
from tests.test_oracle_kats import reference_test_inputs

for f in (br.TWEEDLEDEE_BASE, br.BLS12_377_SCALAR):
    inputs = reference_test_inputs(f.p) + [synth.to_int(r) for r in synth.rand_field(f.field_id, 3, 64)]
    x = ints_to_array(inputs, 4)
    one = ints_to_array([1] * len(inputs), 4)
    got = array_to_ints(api.field_op(f.field_id, "mul", x, one))      # x * 1 / R
    exp = [v * f.Rinv % f.p for v in inputs]
    bad = [i for i in range(len(inputs)) if got[i] != exp[i]]
    print(f.name, "x*1: bad", len(bad), "of", len(inputs))
    for i in bad[:6]:
        print("   in ", hex(inputs[i])); print("   got", hex(got[i])); print("   exp", hex(exp[i])); print("   diff", hex((got[i]-exp[i]) % f.p))
    got = array_to_ints(api.field_op(f.field_id, "square", x))
    exp = [v * v * f.Rinv % f.p for v in inputs]
    bad = [i for i in range(len(inputs)) if got[i] != exp[i]]
    print(f.name, "square: bad", len(bad))
    for i in bad[:4]:
        print("   in ", hex(inputs[i])); print("   got", hex(got[i])); print("   exp", hex(exp[i])); print("   diff", hex((got[i]-exp[i]) % f.p), "got<p", got[i] < f.p)
