"""Development tool: long randomised attack on the message certificate (tests/adversarial.py) --
device messages (stereo_trws_messages) against the reference's type classes where oracle/_ref is
built, the oracle's restatement otherwise.  Prints and optionally writes a JSON summary.
usage: stress_certificate.py [seconds=60] [first_seed=0] [summary.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import adversarial
from oracle import pyoracle as po
from stereo_amd.trws import messages

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
impl = "ref" if po.have_ref_types() else "envelope"
t0, first = time.time(), seed
n = ser = 0
bad = []
while time.time() - t0 < budget:
    kernel = 1 + seed % 2
    K = (5, 16, 33, 48, 60, 64)[(seed // 2) % 6]
    b, s, info = adversarial.check(messages, po, seed, kernel, K, 400, impl=impl)
    n += 400; ser += s
    if b:
        bad.append(dict(seed=seed, kernel=kernel, K=K, rows=b[:10], positions=info["mode"]))
        print("MISMATCH", bad[-1])
    seed += 1
summary = dict(tool="tools/stress_certificate.py", against=impl, messages=n, batches=seed - first, first_seed=first,
               last_seed=seed - 1, took_serial_path=ser, mismatches=len(bad), mismatch_details=bad[:20],
               seconds=round(time.time() - t0, 1))
print("stress certificate: %d messages (%d via the serial path), %d mismatches, %.0f s, against %s" % (
    n, ser, len(bad), time.time() - t0, impl))
if len(sys.argv) > 3:
    json.dump(summary, open(sys.argv[3], "w"), indent=1)
sys.exit(1 if bad else 0)
