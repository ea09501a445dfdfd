"""CPU: the native trainer's parameter table (no device needed: nothing is launched) -- the reference's state-dict keys and
shapes in the reference's order, a flat kernel-layout buffer that holds them all, and workspace sizes that grow with the batch."""
import ctypes

from densefusion_amd import _lib, synth


def _spec(h):
    L = _lib.lib()
    key, shape, ndim = ctypes.create_string_buffer(256), (ctypes.c_int64 * 4)(), ctypes.c_int()
    out = []
    for i in range(L.df_trainer_num_params(h)):
        assert L.df_trainer_param_info(h, i, key, 256, shape, ctypes.byref(ndim)) == 0
        out.append((key.value.decode(), tuple(int(shape[d]) for d in range(ndim.value))))
    return out


def test_trainer_tables_follow_the_reference_state_dicts():
    L = _lib.lib()
    for kind, spec_fn, K, N in ((0, synth.posenet_spec, 21, 1000), (1, synth.refiner_spec, 21, 1000), (0, synth.posenet_spec, 13, 500)):
        h = L.df_trainer_create(kind, N, K)
        assert h
        want = [(k, tuple(s)) for k, s in spec_fn(K)]
        assert _spec(h) == want                                  # 77 / 24 tensors, reference keys, shapes and order
        numel = sum(int(__import__("numpy").prod(s)) for _, s in want)
        flat = L.df_trainer_flat_numel(h)
        assert numel <= flat <= numel + 64 * len(want) + 64 * 1024      # padding: 256-byte slots, the stem's 4th input channel
        if kind == 0:
            a, b = L.df_posenet_train_workspace_bytes(h, 1, 80, 80, 500), L.df_posenet_train_workspace_bytes(h, 8, 160, 160, 500)
            assert 0 < a < b < (8 << 30)
            assert L.df_posenet_train_workspace_bytes(h, 0, 80, 80, 500) == 0
            assert L.df_refiner_train_workspace_bytes(h, 1, 500) == 0          # wrong kind
        else:
            a, b = L.df_refiner_train_workspace_bytes(h, 1, 500), L.df_refiner_train_workspace_bytes(h, 8, 2600)
            assert 0 < a < b
        L.df_trainer_destroy(h)
    assert not L.df_trainer_create(2, 10, 10)
