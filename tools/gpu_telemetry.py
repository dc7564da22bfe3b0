"""Clock / socket-power telemetry for the bench line (measurement infrastructure, not product code).

The f16x2 matrix kernels run at the socket's power cap and the firmware trades clock for utilisation, so the same
binary is 5-14 % apart between boxes of the pool. `Sampler` polls the ROCm SMI library (librocm_smi64, ctypes, no process
spawn) from a background thread while a region runs and reports the mean / min / max shader clock and socket power, so a
bench line explains itself. Everything is best effort: any failure yields {"available": false, "error": ...}.
"""
from __future__ import annotations

import ctypes as C
import threading
import time


class _Freqs(C.Structure):                       # rsmi_frequencies_t (rocm_smi.h): RSMI_MAX_NUM_FREQUENCIES = 33
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


class Sampler:
    def __init__(self, device: int = 0, period_s: float = 0.02):
        self.dev, self.period = device, period_s
        self.lib, self.err = None, None
        self.clk, self.pw, self.mclk, self.temps, self.other = [], [], [], {}, {}
        self._stop = threading.Event()
        self._thr = None
        try:
            self.lib = C.CDLL("librocm_smi64.so")
            rc = self.lib.rsmi_init(C.c_uint64(0))
            if rc != 0:
                raise OSError(f"rsmi_init -> {rc}")
        except OSError as e:
            self.lib, self.err = None, repr(e)

    def _power_w(self):
        v = C.c_uint64(0)
        if hasattr(self.lib, "rsmi_dev_current_socket_power_get") and \
                self.lib.rsmi_dev_current_socket_power_get(C.c_uint32(self.dev), C.byref(v)) == 0 and v.value:
            return v.value / 1e6
        if self.lib.rsmi_dev_power_ave_get(C.c_uint32(self.dev), C.c_uint32(0), C.byref(v)) == 0 and v.value:
            return v.value / 1e6
        return None

    def _sclk_mhz(self):
        f = _Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.dev), C.c_int(0), C.byref(f)) == 0 and f.num_supported:
            return f.frequency[min(f.current, 32)] / 1e6
        return None

    def _clk_mhz(self, kind):
        f = _Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.dev), C.c_int(kind), C.byref(f)) == 0 and f.num_supported:
            return f.frequency[min(f.current, 32)] / 1e6
        return None

    def _mclk_mhz(self):
        return self._clk_mhz(4)                                                                                          # RSMI_CLK_TYPE_MEM

    def _temps_c(self):
        """junction (1), memory (2) and the hottest HBM stack (3 .. 6) in degrees C, whichever sensors the part reports"""
        out = {}
        for name, sensors in (("junction", (1,)), ("memory", (2,)), ("hbm_max", (3, 4, 5, 6))):
            vals = []
            for sn in sensors:
                v = C.c_int64(0)
                if self.lib.rsmi_dev_temp_metric_get(C.c_uint32(self.dev), C.c_uint32(sn), C.c_int(0), C.byref(v)) == 0 and v.value:
                    vals.append(v.value / 1e3)
            if vals:
                out[name] = max(vals)
        return out

    def _run(self):
        k = 0
        while not self._stop.is_set():
            try:
                c, p = self._sclk_mhz(), self._power_w()
                if c:
                    self.clk.append(c)
                if p:
                    self.pw.append(p)
                if k % 10 == 0:                           # memory clock and temperatures: a slow box of the pool is usually a hot
                    m = self._mclk_mhz()                  # or memory-throttled one (memory-bound kernels +60 %)
                    if m:
                        self.mclk.append(m)
                    for name, v in self._temps_c().items():
                        self.temps.setdefault(name, []).append(v)
                    for name, kind in (("df", 1), ("soc", 3)):             # data-fabric and SOC clocks (RSMI_CLK_TYPE_DF / _SOC)
                        v = self._clk_mhz(kind)
                        if v:
                            self.other.setdefault(name, []).append(v)
                k += 1
            except Exception as e:                        # noqa: BLE001  (telemetry must never take the bench down)
                self.err = repr(e)
                return
            self._stop.wait(self.period)

    def start(self):
        self.clk, self.pw, self.mclk, self.temps, self.other = [], [], [], {}, {}
        self._stop.clear()
        if self.lib is not None:
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def stop(self) -> dict:
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=2.0)
            self._thr = None
        if self.lib is None or (not self.clk and not self.pw):
            return {"available": False, "error": self.err or "no samples"}

        def stats(xs, nd):
            return (round(sum(xs) / len(xs), nd), round(min(xs), nd), round(max(xs), nd)) if xs else (None, None, None)
        cm, cl, ch = stats(self.clk, 0)
        pm, pl, ph = stats(self.pw, 1)
        return {"available": True, "source": "librocm_smi64 (rsmi_dev_gpu_clk_freq_get SYS, socket power), polled every "
                                             f"{self.period * 1e3:.0f} ms from a host thread",
                "samples": max(len(self.clk), len(self.pw)), "sclk_mhz_mean": cm, "sclk_mhz_min": cl, "sclk_mhz_max": ch,
                "power_w_mean": pm, "power_w_min": pl, "power_w_max": ph,
                "mclk_mhz_mean": stats(self.mclk, 0)[0], "mclk_mhz_min": stats(self.mclk, 0)[1],
                "temp_c_max": {k: round(max(v), 1) for k, v in self.temps.items()} or None,
                "fabric_soc_clk_mhz_mean": {k: round(sum(v) / len(v), 0) for k, v in self.other.items()} or None}


if __name__ == "__main__":
    s = Sampler().start()
    time.sleep(0.5)
    print(s.stop())
