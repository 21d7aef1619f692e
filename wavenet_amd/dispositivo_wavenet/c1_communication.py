"""Stop-and-wait transfer of strings and files over the modem: the reference's
c1_communication.py (DispositivoWaveNET/dispositivo_wavenet/c1_communication.py) on the
synthetic channel.  Sequencing, retry counts, acknowledgements and return values follow
the reference; waiting is channel time (`canal().sleep`), not wall time.

A sender puts an info frame on air (frame count + name), then the data frames, the last
one re-typed FINAL; after each frame it waits for the receiver's acknowledgement: a ping
tone for strings (:277-297), an OK frame carrying the frame's checksum for files
(:299-327).  Every frame gets TIMES_TO_COMUNICATE_128_BYTES attempts.
"""
from __future__ import annotations

import logging

from .c1_shared import *  # noqa: F401,F403  (the reference re-exports the whole surface this way)
from .c1_shared import (TIME_TO_SAY_128_BYTES, TIMES_TO_COMUNICATE_128_BYTES, TIMES_TO_COMUNICATE_OK, TIPO_OK,
                        TIPO_TRAMA_ARCHIVO, TIPO_TRAMA_ARCHIVO_INFO, TIPO_TRAMA_FINAL_ARCHIVO, Trama, canal,
                        crear_trama_archivo_info, crear_trama_ok, decodificar_payload_archivo_info, ejecutar_ping,
                        emitir_trama, escuchar_ping, escuchar_y_retornar_trama, mac_str_to_bytes,
                        obtener_tramas_desde_archivo, obtener_tramas_desde_string, transmite_freq,
                        verificar_datos_esperados, PING_FREQ)

_ESPERA_TRAMA = TIME_TO_SAY_128_BYTES + 10
_ESCUCHAS_POR_INTENTO = max(TIMES_TO_COMUNICATE_OK - 1, 1)


def _dormir(segundos):
    canal().sleep(segundos)


def _marcar_final(tramas):
    """The last data frame travels as type FINAL, checksum rebuilt (:127-129, :263-265)."""
    if tramas:
        tramas[-1].tipo = TIPO_TRAMA_FINAL_ARCHIVO
        tramas[-1]._build()


def _tipo_esperado(i, total):
    return TIPO_TRAMA_FINAL_ARCHIVO if i == total - 1 else TIPO_TRAMA_ARCHIVO


# ------------------------------------------------------------------------------------------ senders
def emitir_hasta_respuesta_ping(trama, my_origin_bytes, my_sender_bytes, timeout=None):
    """Emit until a ping tone answers (:277-297)."""
    for intento in range(TIMES_TO_COMUNICATE_128_BYTES):
        logging.info(f"Emitiendo trama por {intento + 1} vez")
        emitir_trama(trama)
        _dormir(1)
        for _ in range(_ESCUCHAS_POR_INTENTO):
            if escuchar_ping(3):
                _dormir(1)
                return True
    return False


def emitir_hasta_respuesta(trama, my_origin_bytes, my_sender_bytes, timeout=None):
    """Emit until an OK frame whose payload is this frame's checksum answers (:299-327)."""
    if timeout is None:
        timeout = _ESPERA_TRAMA
    for intento in range(TIMES_TO_COMUNICATE_128_BYTES):
        logging.info(f"Emitiendo trama por {intento + 1} vez")
        emitir_trama(trama)
        for _ in range(_ESCUCHAS_POR_INTENTO):
            try:
                respuesta = escuchar_y_retornar_trama(timeout=timeout)
            except Exception:
                continue
            respuesta.imprimir()
            if (verificar_datos_esperados(respuesta, TIPO_OK, my_sender_bytes, my_origin_bytes)
                    and respuesta.payload == trama.checksum):
                return True
    return False


def enviar_string_por_sonido(string, mac_org_str, mac_dest_str, timeout=None):
    """True once every frame of `string` was acknowledged by a ping (:235-274)."""
    if timeout is None:
        timeout = _ESPERA_TRAMA
    tramas = obtener_tramas_desde_string(string, mac_org_str, mac_dest_str)
    origen, destino = mac_str_to_bytes(mac_org_str), mac_str_to_bytes(mac_dest_str)
    logging.info("Enviando string...")
    info = crear_trama_archivo_info(origen, destino, len(tramas), "str")
    if not emitir_hasta_respuesta_ping(info, origen, destino, timeout=timeout):
        logging.warning("No se logro comunicar la informacion inicial del string")
        return False
    _marcar_final(tramas)
    for i, trama in enumerate(tramas):
        logging.info(f"Comunicando la trama numero {i + 1}...")
        if not emitir_hasta_respuesta_ping(trama, origen, destino, timeout=timeout):
            logging.warning(f"No se comunico correctamente la trama {i + 1}")
            return False
        _dormir(1)
        ejecutar_ping()
    logging.info("String enviado")
    return True


def enviar_archivo_por_sonido(nombre_archivo, mac_org_str, mac_dest_str):
    """True once every frame of the file was acknowledged by an OK frame (:101-140)."""
    tramas = obtener_tramas_desde_archivo(nombre_archivo, mac_org_str, mac_dest_str)
    origen, destino = mac_str_to_bytes(mac_org_str), mac_str_to_bytes(mac_dest_str)
    logging.info(f"Enviando archivo info del archivo: {nombre_archivo}...")
    info = crear_trama_archivo_info(origen, destino, len(tramas), nombre_archivo)
    if not emitir_hasta_respuesta(info, origen, destino):
        logging.warning("No se logro comunicar la informacion inicial del archivo")
        return False
    _marcar_final(tramas)
    for i, trama in enumerate(tramas):
        logging.info(f"Comunicando la trama numero {i + 1}...")
        if not emitir_hasta_respuesta(trama, origen, destino):
            logging.warning(f"No se comunico correctamente la trama {i + 1}")
            return False
        _dormir(0.5)
        for _ in range(5):
            transmite_freq(PING_FREQ)
        _dormir(1)
    logging.info("Archivo enviado")
    return True


# ------------------------------------------------------------------------------------------ receivers
def _escuchar_info(mi_mac, timeout):
    """Up to three listens for an info frame addressed to me; returns it, or -1 / None the
    way the reference's loop leaves its variable (:159-168)."""
    encontrada = None
    for _ in range(TIMES_TO_COMUNICATE_128_BYTES):
        try:
            encontrada = escuchar_y_retornar_trama(timeout=timeout)
            encontrada.imprimir()
            if encontrada.tipo == TIPO_TRAMA_ARCHIVO_INFO and encontrada.mac_destino == mi_mac:
                return encontrada
        except Exception as e:
            logging.warning(f"error: {e}")
        encontrada = -1
    return encontrada


def escuchar_string(my_mac_address_str, timeout=None, init_timeout=None):
    """The received text, or False (:142-232)."""
    mi_mac = mac_str_to_bytes(my_mac_address_str)
    if timeout is None:
        timeout = _ESPERA_TRAMA
    if init_timeout is None:
        init_timeout = 30
    logging.info("Listening...")
    info = _escuchar_info(mi_mac, init_timeout)
    if type(info) is not Trama:
        logging.info("No se escucho ninguna trama con la info del string")
        return False
    if not info.get_checksum_valido():
        logging.warning("Falló el checksum...")
        return False
    try:
        total, _nombre = decodificar_payload_archivo_info(info.payload)
    except Exception as e:
        logging.warning(f"Error al decodificar el payload: {e}")
        return False
    recibido = b""
    ejecutar_ping()
    remitente = info.mac_origen
    try:
        for i in range(total):
            logging.info(f"Trama {i + 1}/{total}")
            trama = None
            for _ in range(TIMES_TO_COMUNICATE_128_BYTES):
                try:
                    trama = escuchar_y_retornar_trama(timeout=timeout)
                    trama.imprimir()
                    if not trama.get_checksum_valido():
                        logging.warning(f"El checksum de la trama {i + 1} es invalido")
                        continue
                    if verificar_datos_esperados(trama, _tipo_esperado(i, total), remitente, mi_mac):
                        break
                except Exception as e:
                    logging.warning(f"Error al recibir trama: {e}")
                    continue
            if trama is None:
                logging.warning(f"No se pudo escuchar la trama {i + 1}")
                return False
            recibido += trama.payload
            _dormir(1)
            ejecutar_ping()
    except Exception as e:
        logging.warning(f"Error al recibir el string: {e}")
        return False
    logging.info(f"{recibido}")
    return recibido.decode()


def escuchar_archivo(my_mac_address_str):
    """Receives a file into new_<name>; True on success (:5-96)."""
    mi_mac = mac_str_to_bytes(my_mac_address_str)
    info = None
    for _ in range(TIMES_TO_COMUNICATE_128_BYTES):
        try:
            info = escuchar_y_retornar_trama(timeout=_ESPERA_TRAMA)
            info.imprimir()
            if info.tipo == TIPO_TRAMA_ARCHIVO_INFO and info.mac_destino == mi_mac:
                break
        except Exception:
            logging.warning("error")
        info = -1
    if info is None:
        logging.info("No se escucho ninguna trama con la info del archivo inicial")
        return False
    info.imprimir()     # after three failed listens `info` is -1 and this raises, as it does in the reference (:27-34)
    try:
        total, nombre = decodificar_payload_archivo_info(info.payload)
    except Exception as e:
        logging.warning(f"Error al decodificar el payload: {e}")
        return False
    ok_info = crear_trama_ok(mi_mac, info.mac_origen, info.checksum)
    nombre = "new_" + nombre
    for _ in range(TIMES_TO_COMUNICATE_OK):
        emitir_trama(ok_info)
    remitente = info.mac_origen
    try:
        with open(nombre, 'wb') as f:
            for i in range(total):
                trama = None
                for _ in range(TIMES_TO_COMUNICATE_128_BYTES):
                    try:
                        trama = escuchar_y_retornar_trama(_ESPERA_TRAMA)
                        trama.imprimir()
                        if verificar_datos_esperados(trama, _tipo_esperado(i, total), remitente, mi_mac):
                            break
                    except Exception:
                        continue
                if trama is None:
                    logging.warning(f"No se pudo escuchar la trama {i + 1}")
                    return False
                if not trama.get_checksum_valido():
                    logging.warning(f"El checksum de la trama {i + 1} es invalido")
                    return False
                f.write(trama.payload)
                ok = crear_trama_ok(mi_mac, trama.mac_origen, trama.checksum)
                for _ in range(TIMES_TO_COMUNICATE_OK):
                    emitir_trama(ok)
                    if escuchar_ping(5):
                        break
    except Exception as e:
        logging.warning(f"Error al escribir el archivo: {e}")
        return False
    logging.info("Archivo recibido exitosamente.")
    return True
