#!/usr/bin/env python
"""Drive tools/probes/pk_opsel_probe.hip: each packed instruction form alone and beside the encoders (second stream)."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dpvo_amd.encoders import HipEncoders
from dpvo_amd.net import VONet

lib = ctypes.CDLL(os.path.join(HERE, "libpk_opsel_probe.so"))
lib.pk_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
torch.manual_seed(0)
vo = VONet().to(dev)
enc = HipEncoders(vo.patchify.fnet, vo.patchify.inet)
img = (torch.randn(3, 480, 640, device=dev) / 2).half()
eo = (torch.empty(120, 160, 128, dtype=torch.float16, device=dev), torch.empty(120, 160, 384, dtype=torch.float16, device=dev))
side = torch.cuda.Stream(device=dev)
out = torch.zeros(20, dtype=torch.int32, device=dev)
FORMS = ["pk_mul op_sel:[0,1] op_sel_hi:[1,0]", "pk_mul op_sel:[1,0] op_sel_hi:[0,1]", "pk_mul (default selects)",
         "pk_mul op_sel_hi:[0,1]", "pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1]", "pk_add op_sel:[0,1] op_sel_hi:[1,0]",
         "pk_mov op_sel:[1,0]", "pk_mul op_sel:[1,1] op_sel_hi:[0,0]", "pk_mul op_sel:[0,1]", "pk_fma op_sel:[0,0,1] op_sel_hi:[1,1,0]",
         "pk_fma op_sel:[1,0,0] op_sel_hi:[0,1,1]", "pk_fma op_sel:[0,1,1] op_sel_hi:[1,0,0]", "pk_fma op_sel:[1,1,0] op_sel_hi:[0,0,1]"]
iters, blocks, reps = int(os.environ.get("ITERS", "20000")), int(os.environ.get("BLOCKS", "1024")), int(os.environ.get("REPS", "30"))
mm = torch.randn(4096, 4096, device=dev, dtype=torch.half)
lib.pk_neighbour_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
KINDS = ["v_fma_mix_f32 (op_sel)", "v_cvt_f32_f16_sdwa WORD_1", "ds_bpermute_b32", "v_pk_max_f16 (op_sel)", "v_fma_f64", "v_mfma_f32_16x16x32_f16",
         "v_perm_b32", "v_pk_mul_f32", "v_fma_f32", "v_accvgpr write/read", "v_add_f32_sdwa WORD_1", "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]",
         "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_32x32x8_f16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x32_bf16",
         "v_mfma_f32_16x16x32_f16 (VGPR accumulator)", "v_mfma_f64_16x16x4_f64"]
sink = torch.zeros(256, device=dev)
import struct
f = lambda u: struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
if os.environ.get("SYNTH", "1") == "1":
    for kind, kname in enumerate(KINDS):
        tot = torch.zeros(20, dtype=torch.int64); sample = None
        for r in range(reps):
            with torch.cuda.stream(side):
                assert lib.pk_neighbour_launch(ctypes.c_void_p(side.cuda_stream), kind, int(os.environ.get("NITERS", "6000")), 2048, ctypes.c_void_p(sink.data_ptr())) == 0
            assert lib.pk_probe_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0, iters, blocks, ctypes.c_void_p(out.data_ptr()), 0) == 0
            o = out.cpu().long(); tot += o
            if sample is None and int(o[12]):
                sample = o[13:20].tolist()
            torch.cuda.synchronize()
        print(f"synthetic neighbour {kname:44s} victim pk_mul op_sel:[0,1] op_sel_hi:[1,0]: low-result faults by lane quarter {tot[:4].tolist()}", flush=True)
        if sample:
            print(f"    first fault: lane {sample[0]} iteration {sample[1]} got {f(sample[2])} with a = ({f(sample[3])}, {f(sample[4])}) b = ({f(sample[5])}, {f(sample[6])}); expected a.lo*b.hi = {f(sample[3]) * f(sample[6])}")
if os.environ.get("TABLE", "1") != "1":
    FORMS = FORMS[:1]
for neighbour in ("none", "matmul", "encoders", "mfma16x16x32"):
    for nop in (0,):
        for form, name in enumerate(FORMS):
            tot = torch.zeros(12, dtype=torch.int64)
            for r in range(reps):
                if neighbour == "mfma16x16x32":
                    with torch.cuda.stream(side):
                        assert lib.pk_neighbour_launch(ctypes.c_void_p(side.cuda_stream), 5, int(os.environ.get("NITERS", "6000")), 2048, ctypes.c_void_p(sink.data_ptr())) == 0
                elif neighbour != "none":
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            enc(img, fmap_out=eo[0], imap_out=eo[1]) if neighbour == "encoders" else torch.mm(mm, mm)
                rc = lib.pk_probe_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), form, iters, blocks, ctypes.c_void_p(out.data_ptr()), nop)
                assert rc == 0, rc
                o = out.cpu().long(); tot += o[:12]
                if int(o[12]) and r == 0:
                    sm = o[13:20].tolist()
                    print(f"    first fault: lane {sm[0]} iteration {sm[1]} got {f(sm[2])} with a = ({f(sm[3])}, {f(sm[4])}) b = ({f(sm[5])}, {f(sm[6])})")
            torch.cuda.synchronize()
            n = reps * blocks * iters
            print(f"neighbour {neighbour:8s} s_nop {nop}  {name:42s} wave-instructions {n:.2e}  low-result faults by lane quarter {tot[:4].tolist()}"
                  f"  high {tot[4:8].tolist()}  [= unselected half {int(tot[8])}, = previous result {int(tot[9])}, other {int(tot[10])}]", flush=True)
