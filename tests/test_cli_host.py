"""The command line of dispositivo_wavenet (reference dispositivo_wavenet.py:54-98): the parts
that need no GPU."""
import pytest

from wavenet_amd.dispositivo_wavenet import dispositivo_wavenet as dw

MAC = "aa:bb:cc:dd:ee:ff"


@pytest.mark.parametrize("modo", ["1", "3"])
def test_missing_file_is_reported_and_nothing_runs(modo, capsys, tmp_path):
    ruta = str(tmp_path / "no_existe.txt")
    dw.main(["-a", ruta, "-b", MAC, "-c", MAC, "-d", modo])
    assert capsys.readouterr().out.strip() == f"El archivo {ruta} no existe."


def test_unknown_mode_does_nothing(capsys):
    dw.main(["-a", "x", "-b", MAC, "-d", "9"])
    assert capsys.readouterr().out == ""


def test_required_options(capsys):
    with pytest.raises(SystemExit):
        dw.main(["-a", "x", "-d", "2"])          # -b is required
    assert "mac_origen" in capsys.readouterr().err


def test_device_object_keeps_the_reference_surface():
    d = dw.DispositivoWaveNet(MAC)
    assert (d.mac_origen, d.mac_destino) == (MAC, None)
    assert d.ERROR_ENVIO == "Failed to send string through layer 1"
    assert d.ERROR_ESCUCHA == "Failed to get string from layer 1"
    # the module re-exports layer 1 the way the reference module does
    for nombre in ("enviar_string_por_sonido", "escuchar_string", "Trama", "calcular_checksum", "TIME_TO_SAY_128_BYTES"):
        assert hasattr(dw, nombre)
