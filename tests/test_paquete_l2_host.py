"""The layer-2 packet text helper of the sound-link test (tests/paquete_l2.py): behaviours read
from WaveNetPacketeering.py:152-193.  Parity unpinned (see the helper's docstring)."""
import json

from paquete_l2 import Paquete, PaqueteSecreto, es_nulo, forma, reconstruir


def test_round_trip_and_key_order():
    p = Paquete(1, 2, "msg", "ahu, ahua", "nope")          # the packet of WaveNetCore/tests/test_sound_protocol_send.py:8
    t = forma(p)
    assert t == '{"enc": false, "src": 1, "dest": 2, "mtype": "msg", "body": "ahu, ahua", "timestamp": "nope"}'
    assert reconstruir(t) == p and not es_nulo(reconstruir(t))


def test_malformed_texts_give_the_null_packet():
    base = {"enc": False, "src": 1, "dest": 2, "mtype": "msg", "body": "x", "timestamp": "t"}
    for quitar in ("enc", "src", "dest", "mtype", "body", "timestamp"):
        d = {k: v for k, v in base.items() if k != quitar}
        r = reconstruir(json.dumps(d))
        assert es_nulo(r) and r.body == f"Missing {quitar} tag" and (r.src, r.dest) == (-1, -1)
    for malo, valor in (("src", "1"), ("dest", 2.0), ("src", True), ("mtype", 3), ("body", None), ("enc", 0)):
        r = reconstruir(json.dumps(dict(base, **{malo: valor})))
        assert es_nulo(r) and r.body == f"Bad {malo} tag"
    r = reconstruir('{"enc": false, "src": 1')
    assert es_nulo(r) and r.body.startswith("Formation Error ")
    r = reconstruir("[1, 2]")                               # JSON, but not an object holding "enc"
    assert es_nulo(r) and r.body == "Missing enc tag"
    assert es_nulo(reconstruir(None)) and reconstruir(None).body.startswith("Formation Error ")


def test_encrypted_form_is_recognised():
    r = reconstruir(json.dumps({"enc": True, "meta": "bWV0YQ==", "body": "Y3VlcnBv"}))
    assert r == PaqueteSecreto("bWV0YQ==", "Y3VlcnBv")
    assert reconstruir(json.dumps({"enc": True, "meta": "x"})).body == "Missing body tag"
