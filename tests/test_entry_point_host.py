"""Host-side behaviour of the entry point that needs no GPU: config validation (video_processing.py:87-98 and this build's
added keys in the same style), argument errors raised before any device work, and the CSV writer under threads (:44-68)."""
import csv
import threading

import numpy as np
import pytest

from rtvqa_amd import video_processing as vp

GOOD = {"crf": 23, "vmaf_model_path": None, "resize_width": 64, "resize_height": 64, "frame_interval": 10}


def test_reference_config_validates_and_keeps_its_messages():
    vp.validate_config(dict(GOOD))
    for bad, msg in (({"crf": 0}, "CRF value must be between 1 and 51."), ({"resize_width": 0}, "Resize dimensions must be positive integers."),
                     ({"frame_interval": 0}, "Frame interval must be a positive integer."), ({"num_workers": "4"}, "num_workers must be an integer.")):
        with pytest.raises(ValueError, match=msg.replace(".", r"\.")):
            vp.validate_config(dict(GOOD, **bad))


@pytest.mark.parametrize("key,good,bad,msg", [
    ("ssim_mode", ("gauss", "ffmpeg"), ("ms-ssim", None, 1), "ssim_mode must be 'gauss' or 'ffmpeg'."),
    ("pixfmt", (None, "bgr24", "yuv420p", "gray"), ("nv12", 0), "pixfmt must be 'bgr24', 'yuv420p' or 'gray'."),
    ("dct_mode", (None, "auto", "block8", "full"), ("8x8", 8), "dct_mode must be 'auto', 'block8' or 'full'."),
    ("motion", (None, "sad", "farneback"), ("flow", 1), "motion must be 'sad' or 'farneback'."),
    ("device", (None, 0, 7), (-1, "0", 1.0, True), "device must be a non-negative integer."),
    ("batch_size", (1, 100), (0, -5, "100", 2.5), "batch_size must be a positive integer."),
])
def test_added_config_keys_are_validated_in_the_reference_style(key, good, bad, msg):
    for v in good:
        vp.validate_config(dict(GOOD, **{key: v}))
    for v in bad:
        with pytest.raises(ValueError) as e:
            vp.validate_config(dict(GOOD, **{key: v}))
        assert str(e.value) == msg
        # the entry point refuses the same values before it opens a stream or touches a device
        with pytest.raises(ValueError) as e:
            vp.process_video_and_extract_metrics(np.zeros((2, 8, 8, 3), np.uint8), np.zeros((2, 8, 8, 3), np.uint8), dict(GOOD, **{key: v}))
        assert str(e.value) == msg


def test_load_config_accepts_the_added_keys(tmp_path):
    import json
    p = tmp_path / "config.json"
    p.write_text(json.dumps(dict(GOOD, ssim_mode="ffmpeg", pixfmt="yuv420p", dct_mode="full", motion="farneback", device=0, batch_size=64)))
    assert vp.load_config(str(p))["motion"] == "farneback"
    p.write_text(json.dumps(dict(GOOD, motion="optical")))
    with pytest.raises(ValueError, match="motion must be"):
        vp.load_config(str(p))


def test_planar_pair_without_the_bgr_stream_is_refused_before_any_device_work(tmp_path):
    from rtvqa_amd import frames
    h, w = 16, 24
    y = np.zeros((3, frames.frame_bytes_yuv420p(h, w)), np.uint8)
    pr = str(tmp_path / "r.y4m")
    frames.write_y4m(pr, y, h, w)
    with pytest.raises(ValueError, match="encoded_bgr"):
        vp.process_video_and_extract_metrics(pr, pr, dict(GOOD))
    with pytest.raises(ValueError, match="encoded_bgr"):
        vp.process_video_and_extract_metrics(y, y, dict(GOOD, pixfmt="yuv420p"), height=h, width=w)
    with pytest.raises(ValueError, match="planar"):
        vp.process_video_and_extract_metrics(pr, np.zeros((3, h, w, 3), np.uint8), dict(GOOD), encoded_bgr=np.zeros((3, h, w, 3), np.uint8))
    with pytest.raises(ValueError, match="pixel layout"):
        vp.process_video_and_extract_metrics(np.zeros((3, h, w, 3), np.uint8), pr, dict(GOOD), encoded_bgr=np.zeros((3, h, w, 3), np.uint8))
    with pytest.raises(ValueError, match="Unsupported file type"):
        vp.process_video_and_extract_metrics("a.mp4", "b.mp4", dict(GOOD))
    with pytest.raises(ValueError, match="needs height and width"):     # headerless planes: the geometry must be named
        vp.process_video_and_extract_metrics("a.yuv", "b.yuv", dict(GOOD, pixfmt="yuv420p"), encoded_bgr=np.zeros((3, h, w, 3), np.uint8))
    with pytest.raises(ValueError, match="Unsupported file type"):
        vp.process_video_and_extract_metrics("a.nv12", "b.nv12", dict(GOOD, pixfmt="yuv420p"), encoded_bgr=np.zeros((3, h, w, 3), np.uint8))


def test_csv_writer_under_threads_writes_one_header(tmp_path):
    """the reference tests for the file outside its lock (video_processing.py:56 vs :62): here the test is inside"""
    out = str(tmp_path / "rows.csv")
    start = threading.Barrier(16)

    def work(k):
        start.wait()
        for i in range(25):
            vp.thread_safe_update_csv({"a": k, "b": i, "c": "x"}, out)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(16)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    rows = list(csv.reader(open(out)))
    assert rows[0] == ["a", "b", "c"] and len(rows) == 1 + 16 * 25
    assert sum(r == ["a", "b", "c"] for r in rows) == 1
    assert sorted((int(r[0]), int(r[1])) for r in rows[1:]) == [(k, i) for k in range(16) for i in range(25)]


def test_pass_lock_is_one_reentrant_lock_per_device():
    from rtvqa_amd import stream
    a, b = stream.pass_lock(0), stream.pass_lock(1)
    assert a is stream.pass_lock(0) and a is not b
    with a:
        with stream.pass_lock(0):     # re-entrant: the entry point holds it around process_in_batches' per-item fallbacks
            held = []
            t = threading.Thread(target=lambda: held.append(a.acquire(timeout=0.05)))
            t.start()
            t.join()
            assert held == [False]    # another thread waits


def test_a_half_built_pinned_ring_is_given_back_whole():
    """_Staging.ring() leaked the slots it had allocated (and kept slot_bytes bumped) when a later
    alloc_pinned failed.  With a stand-in engine whose third allocation fails: everything allocated is freed, the ring is
    empty, and the next call starts from scratch."""
    from rtvqa_amd import stream

    class FakeEngine:
        device = 0

        def __init__(self):
            self.live, self.calls, self.fail_at = [], 0, 3

        def alloc_pinned(self, shape):
            self.calls += 1
            if self.calls == self.fail_at:
                raise MemoryError("pinned allocation failed")
            a = np.zeros(shape, np.uint8)
            self.live.append(id(a))
            return a

        def free_pinned(self, a):
            self.live.remove(id(a))

    eng = FakeEngine()
    st = stream._Staging(eng)
    with pytest.raises(MemoryError):
        st.ring(4, 1000)
    assert eng.live == [] and st.slots == [] and st.slot_bytes == 0
    eng.fail_at = -1
    slots = st.ring(4, 1000)
    assert len(slots) == 4 and len(eng.live) == 4 and st.slot_bytes == 1000
    assert st.ring(2, 500)[0] is slots[0] and len(eng.live) == 4          # big enough: kept
    bigger = st.ring(3, 2000)                                              # a bigger chunk: the old slots go, new ones come
    assert len(bigger) == 3 and len(eng.live) == 3 and st.slot_bytes == 2000
