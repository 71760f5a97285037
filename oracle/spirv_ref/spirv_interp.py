"""A small SPIR-V interpreter: executes the reference's committed shader binaries on the CPU.

TEST INFRASTRUCTURE (oracle pinning).  The reference (expenses/transmission-renderer) has no tests or golden
vectors and its Rust/rust-gpu sources cannot be built here, but it commits the build output of its `shader`
crate under compiled-shaders/*.spv.  This module executes those binaries word by word, one invocation at a
time, in IEEE fp32 (numpy float32 scalars, one rounding per SPIR-V arithmetic instruction — the modules contain
no fused ops), so the arithmetic, the operation order, the control flow and the buffer layouts (explicit
Offset / ArrayStride / MatrixStride decorations, read from raw bytes) are the reference's own.  What a Vulkan
implementation does in fixed function — texel filtering (OpImageSample*), derivatives — is supplied by the
caller through callbacks (tools/make_golden_spirv.py passes the oracle's restatement of the Vulkan rules).

It is NOT a general SPIR-V VM: it covers the opcodes the reference's 15 non-ray-tracing modules use
(logical addressing, structured control flow, GLSL.std.450 subset) and raises on anything else.
The .spv files are read from /root/reference at fixture-generation time only; nothing is copied into this
repository and nothing here runs on the GPU box.
"""
from __future__ import annotations

import struct
from typing import Callable, Dict, List, Optional

import numpy as np

f32 = np.float32
u32 = np.uint32

# opcodes
OP = dict(
    Undef=1, Name=5, MemberName=6, ExtInstImport=11, ExtInst=12, MemoryModel=14, EntryPoint=15, ExecutionMode=16,
    Capability=17, TypeVoid=19, TypeBool=20, TypeInt=21, TypeFloat=22, TypeVector=23, TypeMatrix=24, TypeImage=25,
    TypeSampler=26, TypeSampledImage=27, TypeArray=28, TypeRuntimeArray=29, TypeStruct=30, TypePointer=32,
    TypeFunction=33, ConstantTrue=41, ConstantFalse=42, Constant=43, ConstantComposite=44, ConstantNull=46,
    Function=54, FunctionParameter=55, FunctionEnd=56, FunctionCall=57, Variable=59, Load=61, Store=62,
    AccessChain=65, InBoundsAccessChain=66, ArrayLength=68, Decorate=71, MemberDecorate=72, VectorShuffle=79,
    CompositeConstruct=80, CompositeExtract=81, CompositeInsert=82, CopyObject=83, SampledImage=86,
    ImageSampleImplicitLod=87, ImageSampleExplicitLod=88, ConvertFToU=109, ConvertFToS=110, ConvertSToF=111,
    ConvertUToF=112, Bitcast=124, SNegate=126, FNegate=127, IAdd=128, FAdd=129, ISub=130, FSub=131, IMul=132,
    FMul=133, UDiv=134, SDiv=135, FDiv=136, UMod=137, VectorTimesScalar=142, MatrixTimesVector=145, Dot=148,
    LogicalEqual=164, LogicalNotEqual=165, LogicalOr=166, LogicalAnd=167, LogicalNot=168, Select=169, IEqual=170, INotEqual=171, UGreaterThan=172,
    SGreaterThan=173, UGreaterThanEqual=174, SGreaterThanEqual=175, ULessThan=176, SLessThan=177,
    ULessThanEqual=178, SLessThanEqual=179, FOrdEqual=180, FUnordEqual=181, FOrdNotEqual=182, FUnordNotEqual=183,
    FOrdLessThan=184, FUnordLessThan=185, FOrdGreaterThan=186, FUnordGreaterThan=187, FOrdLessThanEqual=188,
    FUnordLessThanEqual=189, FOrdGreaterThanEqual=190, FUnordGreaterThanEqual=191, ShiftRightLogical=194,
    ShiftLeftLogical=196, BitwiseOr=197, BitwiseXor=198, BitwiseAnd=199, Not=200, DPdx=207, DPdy=208,
    AtomicIIncrement=232, AtomicIAdd=234, Phi=245, LoopMerge=246, SelectionMerge=247, Label=248, Branch=249,
    BranchConditional=250, Switch=251, Kill=252, Return=253, ReturnValue=254, Unreachable=255, IsNan=156, IsInf=157,
)
DEC = dict(Block=2, BufferBlock=3, ArrayStride=6, MatrixStride=7, BuiltIn=11, Flat=14, Location=30, Binding=33,
           DescriptorSet=34, Offset=35)
BUILTIN = dict(Position=0, VertexIndex=42, InstanceIndex=43, FragCoord=15, GlobalInvocationId=28)
SC = dict(UniformConstant=0, Input=1, Uniform=2, Output=3, Workgroup=4, Private=6, Function=7, PushConstant=9,
          StorageBuffer=12)
LAID_OUT = (SC["Uniform"], SC["PushConstant"], SC["StorageBuffer"])


class Discard(Exception):
    """OpKill."""


class Module:
    def __init__(self, path: str):
        data = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(data) // 4), data)
        assert w[0] == 0x07230203, "not SPIR-V"
        self.version, self.bound = w[1], w[3]
        self.types: Dict[int, tuple] = {}
        self.consts: Dict[int, object] = {}
        self.decor: Dict[int, Dict[int, tuple]] = {}
        self.mdecor: Dict[int, Dict[int, Dict[int, tuple]]] = {}
        self.names: Dict[int, str] = {}
        self.vars: Dict[int, tuple] = {}      # id -> (pointer type id, storage class)
        self.entry_points: Dict[str, tuple] = {}
        self.functions: Dict[int, dict] = {}
        self.ext_imports: Dict[int, str] = {}
        i, cur = 5, None
        while i < len(w):
            op, n = w[i] & 0xFFFF, w[i] >> 16
            a = w[i + 1:i + n]
            i += n
            if cur is not None:
                if op == OP["FunctionEnd"]:
                    cur = None
                    continue
                if op == OP["FunctionParameter"]:
                    cur["params"].append(a[1])
                    continue
                if op == OP["Label"]:
                    cur["labels"][a[0]] = len(cur["code"])
                cur["code"].append((op, a))
                continue
            if op == OP["Name"]:
                self.names[a[0]] = self._str(a[1:])
            elif op == OP["ExtInstImport"]:
                self.ext_imports[a[0]] = self._str(a[1:])
            elif op == OP["EntryPoint"]:
                name = self._str(a[2:])
                nwords = len(name) // 4 + 1
                self.entry_points[name] = (a[0], a[1], list(a[2 + nwords:]))
            elif op == OP["Decorate"]:
                self.decor.setdefault(a[0], {})[a[1]] = tuple(a[2:])
            elif op == OP["MemberDecorate"]:
                self.mdecor.setdefault(a[0], {}).setdefault(a[1], {})[a[2]] = tuple(a[3:])
            elif op == OP["TypeVoid"]:
                self.types[a[0]] = ("void",)
            elif op == OP["TypeBool"]:
                self.types[a[0]] = ("bool",)
            elif op == OP["TypeInt"]:
                self.types[a[0]] = ("int", a[1], a[2])
            elif op == OP["TypeFloat"]:
                self.types[a[0]] = ("float", a[1])
            elif op == OP["TypeVector"]:
                self.types[a[0]] = ("vector", a[1], a[2])
            elif op == OP["TypeMatrix"]:
                self.types[a[0]] = ("matrix", a[1], a[2])
            elif op == OP["TypeImage"]:
                self.types[a[0]] = ("image",) + tuple(a[1:])
            elif op == OP["TypeSampler"]:
                self.types[a[0]] = ("sampler",)
            elif op == OP["TypeSampledImage"]:
                self.types[a[0]] = ("sampled_image", a[1])
            elif op == OP["TypeArray"]:
                self.types[a[0]] = ("array", a[1], a[2])  # length is a constant id
            elif op == OP["TypeRuntimeArray"]:
                self.types[a[0]] = ("runtime_array", a[1])
            elif op == OP["TypeStruct"]:
                self.types[a[0]] = ("struct", tuple(a[1:]))
            elif op == OP["TypePointer"]:
                self.types[a[0]] = ("pointer", a[2], a[1])  # (pointee type, storage class)
            elif op == OP["TypeFunction"]:
                self.types[a[0]] = ("function", a[1], tuple(a[2:]))
            elif op in (OP["ConstantTrue"], OP["ConstantFalse"]):
                self.consts[a[1]] = op == OP["ConstantTrue"]
            elif op == OP["Constant"]:
                t = self.types[a[0]]
                if t[0] == "float":
                    assert t[1] == 32
                    self.consts[a[1]] = f32(struct.unpack("<f", struct.pack("<I", a[2]))[0])
                elif t[0] == "int":
                    if t[1] == 64:
                        self.consts[a[1]] = np.uint64(a[2] | (a[3] << 32))
                    else:
                        self.consts[a[1]] = np.int32(struct.unpack("<i", struct.pack("<I", a[2]))[0]) if t[2] else u32(a[2])
            elif op == OP["ConstantComposite"]:
                self.consts[a[1]] = [self.consts[c] for c in a[2:]]
            elif op == OP["ConstantNull"]:
                self.consts[a[1]] = self.zero(a[0])
            elif op == OP["Undef"]:
                self.consts[a[1]] = self.zero(a[0])
            elif op == OP["Variable"]:
                self.vars[a[1]] = (a[0], a[2])
            elif op == OP["Function"]:
                cur = {"result_type": a[0], "params": [], "code": [], "labels": {}}
                self.functions[a[1]] = cur

    @staticmethod
    def _str(words) -> str:
        b = b"".join(struct.pack("<I", x) for x in words)
        return b.split(b"\0", 1)[0].decode()

    def zero(self, tid):
        t = self.types[tid]
        k = t[0]
        if k == "float":
            return f32(0)
        if k == "int":
            return np.int32(0) if t[2] else u32(0)
        if k == "bool":
            return False
        if k == "vector":
            return [self.zero(t[1]) for _ in range(t[2])]
        if k == "matrix":
            return [self.zero(t[1]) for _ in range(t[2])]
        if k == "array":
            return [self.zero(t[1]) for _ in range(int(self.consts[t[2]]))]
        if k == "struct":
            return [self.zero(m) for m in t[1]]
        return None

    # ---- explicit layout -------------------------------------------------------------------
    def size_of(self, tid, matrix_stride=None) -> int:
        t = self.types[tid]
        k = t[0]
        if k in ("float", "int"):
            return t[1] // 8
        if k == "vector":
            return self.size_of(t[1]) * t[2]
        if k == "matrix":
            return (matrix_stride or 16) * t[2]
        if k == "array":
            return self.decor[tid][DEC["ArrayStride"]][0] * int(self.consts[t[2]])
        if k == "struct":
            end = 0
            for i, m in enumerate(t[1]):
                md = self.mdecor.get(tid, {}).get(i, {})
                off = md[DEC["Offset"]][0]
                end = max(end, off + self.size_of(m, md.get(DEC["MatrixStride"], (None,))[0]))
            return end
        raise NotImplementedError(k)

    def read(self, buf: bytes, off: int, tid, matrix_stride=None):
        t = self.types[tid]
        k = t[0]
        if k == "float":
            if off + 4 > len(buf):
                return f32(0)      # robustBufferAccess-style
            return f32(struct.unpack_from("<f", buf, off)[0])
        if k == "int":
            if off + t[1] // 8 > len(buf):
                return u32(0) if not t[2] else np.int32(0)
            if t[1] == 64:
                return np.uint64(struct.unpack_from("<Q", buf, off)[0])
            return np.int32(struct.unpack_from("<i", buf, off)[0]) if t[2] else u32(struct.unpack_from("<I", buf, off)[0])
        if k == "vector":
            s = self.size_of(t[1])
            return [self.read(buf, off + i * s, t[1]) for i in range(t[2])]
        if k == "matrix":
            return [self.read(buf, off + c * (matrix_stride or 16), t[1]) for c in range(t[2])]
        if k == "array":
            st = self.decor[tid][DEC["ArrayStride"]][0]
            return [self.read(buf, off + i * st, t[1]) for i in range(int(self.consts[t[2]]))]
        if k == "struct":
            out = []
            for i, m in enumerate(t[1]):
                md = self.mdecor.get(tid, {}).get(i, {})
                out.append(self.read(buf, off + md[DEC["Offset"]][0], m, md.get(DEC["MatrixStride"], (None,))[0]))
            return out
        raise NotImplementedError(k)

    def write(self, buf: bytearray, off: int, tid, value, matrix_stride=None):
        t = self.types[tid]
        k = t[0]
        if k == "float":
            struct.pack_into("<f", buf, off, float(value))
        elif k == "int":
            struct.pack_into("<i" if t[2] else "<I", buf, off, int(value))
        elif k == "vector":
            s = self.size_of(t[1])
            for i in range(t[2]):
                self.write(buf, off + i * s, t[1], value[i])
        elif k == "array":
            st = self.decor[tid][DEC["ArrayStride"]][0]
            for i in range(int(self.consts[t[2]])):
                self.write(buf, off + i * st, t[1], value[i])
        elif k == "struct":
            for i, m in enumerate(t[1]):
                md = self.mdecor.get(tid, {}).get(i, {})
                self.write(buf, off + md[DEC["Offset"]][0], m, value[i], md.get(DEC["MatrixStride"], (None,))[0])
        else:
            raise NotImplementedError(k)


def _map(fn, *xs):
    if isinstance(xs[0], list):
        return [_map(fn, *[x[i] if isinstance(x, list) else x for x in xs]) for i in range(len(xs[0]))]
    return fn(*xs)


def _f_to_u(x) -> np.uint32:   # Rust `as u32` (what rust-gpu means by OpConvertFToU): saturating, NaN -> 0
    x = float(x)
    if not x > 0.0:
        return u32(0)
    if x >= 4294967296.0:
        return u32(0xFFFFFFFF)
    return u32(int(x))


_old = np.seterr(all="ignore")


class Interp:
    """One invocation of one entry point.

    buffers      {(set, binding): bytes-like}     storage / uniform buffers (bytearray if written)
    push         bytes                             push-constant block
    inputs       {location or builtin name: value} Input variables
    sample       callable(kind, handle, coord, lod) -> [4 floats]; handle = (set, binding, index)
    derivative   callable(op, value) -> value      for OpDPdx/OpDPdy (only normal mapping uses them)
    """

    def __init__(self, module: Module, entry: str, buffers: Dict[tuple, bytes], push: bytes = b"",
                 inputs: Optional[dict] = None, sample: Optional[Callable] = None,
                 derivative: Optional[Callable] = None):
        self.m = module
        self.fn_id = module.entry_points[entry][1]
        self.buffers, self.push = buffers, push
        self.inputs = inputs or {}
        self.sample_cb, self.deriv_cb = sample, derivative
        self.outputs: Dict[object, object] = {}
        self.store: Dict[int, object] = {}     # Private/Function/Input/Output variable contents
        self.steps = 0
        for vid, (ptype, sc) in module.vars.items():
            pointee = module.types[ptype][1]
            if sc == SC["Input"]:
                d = module.decor.get(vid, {})
                if DEC["BuiltIn"] in d:
                    key = {v: k for k, v in BUILTIN.items()}.get(d[DEC["BuiltIn"]][0], d[DEC["BuiltIn"]][0])
                else:
                    key = d[DEC["Location"]][0]
                if key in self.inputs:
                    self.store[vid] = self._coerce(pointee, self.inputs[key])
                else:
                    self.store[vid] = module.zero(pointee)
            elif sc in (SC["Output"], SC["Private"]):
                self.store[vid] = module.zero(pointee)

    def _coerce(self, tid, v):
        t = self.m.types[tid]
        if t[0] == "float":
            return f32(v)
        if t[0] == "int":
            return np.int32(v) if t[2] else u32(v)
        if t[0] in ("vector", "array", "matrix"):
            return [self._coerce(t[1], x) for x in v]
        return v

    # ---- pointers: ("buf", key, byte offset, type, matrix stride) | ("obj", var id, path) | ("img", ...)
    def _var_pointer(self, vid):
        ptype, sc = self.m.vars[vid]
        pointee = self.m.types[ptype][1]
        d = self.m.decor.get(vid, {})
        if sc in LAID_OUT:
            key = "push" if sc == SC["PushConstant"] else (d[DEC["DescriptorSet"]][0], d[DEC["Binding"]][0])
            return ("buf", key, 0, pointee, None)
        if sc == SC["UniformConstant"]:
            return ("img", (d[DEC["DescriptorSet"]][0], d[DEC["Binding"]][0]), None, pointee)
        return ("obj", vid, ())

    def _buf(self, key):
        return self.push if key == "push" else self.buffers[key]

    def _access(self, ptr, indices):
        m = self.m
        if ptr[0] == "buf":
            _, key, off, tid, ms = ptr
            for idx in indices:
                t = m.types[tid]
                idx = int(idx)
                if t[0] == "struct":
                    md = m.mdecor.get(tid, {}).get(idx, {})
                    off += md[DEC["Offset"]][0]
                    ms = md.get(DEC["MatrixStride"], (None,))[0]
                    tid = t[1][idx]
                elif t[0] in ("array", "runtime_array"):
                    off += idx * m.decor[tid][DEC["ArrayStride"]][0]
                    tid = t[1]
                elif t[0] == "vector":
                    off += idx * m.size_of(t[1])
                    tid = t[1]
                elif t[0] == "matrix":
                    off += idx * (ms or 16)
                    tid = t[1]
                else:
                    raise NotImplementedError(t)
            return ("buf", key, off, tid, ms)
        if ptr[0] == "img":
            _, key, index, tid = ptr
            t = m.types[tid]
            assert t[0] in ("runtime_array", "array") and len(indices) == 1
            return ("img", key, int(indices[0]), t[1])
        return ("obj", ptr[1], ptr[2] + tuple(int(i) for i in indices))

    def _load(self, ptr):
        if ptr[0] == "buf":
            return self.m.read(self._buf(ptr[1]), ptr[2], ptr[3], ptr[4])
        if ptr[0] == "img":
            return ("handle", ptr[1], ptr[2] or 0)
        v = self.store[ptr[1]]
        for i in ptr[2]:
            v = v[i]
        return v

    def _store(self, ptr, value):
        if ptr[0] == "buf":
            self.m.write(self._buf(ptr[1]), ptr[2], ptr[3], value, ptr[4])
            return
        if not ptr[2]:
            self.store[ptr[1]] = value
            return
        v = self.store[ptr[1]]
        for i in ptr[2][:-1]:
            v = v[i]
        v[ptr[2][-1]] = value

    # ---- execution ----------------------------------------------------------------------------
    def run(self):
        try:
            self._call(self.fn_id, [])
        except Discard:
            self.outputs["discard"] = True
        for vid, (ptype, sc) in self.m.vars.items():
            if sc == SC["Output"]:
                d = self.m.decor.get(vid, {})
                key = ("builtin", d[DEC["BuiltIn"]][0]) if DEC["BuiltIn"] in d else d.get(DEC["Location"], (vid,))[0]
                self.outputs[key] = self.store[vid]
        return self.outputs

    def _call(self, fn_id, args):
        m = self.m
        fn = m.functions[fn_id]
        code, labels = fn["code"], fn["labels"]
        R: Dict[int, object] = {}
        for pid, a in zip(fn["params"], args):
            R[pid] = a

        def val(i):
            if i in R:
                return R[i]
            if i in m.consts:
                return m.consts[i]
            if i in m.vars:
                return self._var_pointer(i)
            raise KeyError(i)

        pc, prev_label, cur_label = 0, None, None
        while True:
            op, a = code[pc]
            pc += 1
            self.steps += 1
            if op == OP["Label"]:
                prev_label, cur_label = cur_label, a[0]
                # evaluate the block's phis together, from the predecessor
                phis = []
                while code[pc][0] == OP["Phi"]:
                    pa = code[pc][1]
                    for k in range(2, len(pa), 2):
                        if pa[k + 1] == prev_label:
                            phis.append((pa[1], val(pa[k])))
                            break
                    else:
                        raise RuntimeError("phi without matching predecessor")
                    pc += 1
                for rid, v in phis:
                    R[rid] = v
            elif op in (OP["LoopMerge"], OP["SelectionMerge"]):
                pass
            elif op == OP["Branch"]:
                pc = labels[a[0]]
            elif op == OP["BranchConditional"]:
                pc = labels[a[1] if bool(val(a[0])) else a[2]]
            elif op == OP["Switch"]:   # selector, default, (literal, label)*  (32-bit selectors)
                sel = int(val(a[0])) & 0xFFFFFFFF
                target = a[1]
                for k in range(2, len(a), 2):
                    if (a[k] & 0xFFFFFFFF) == sel:
                        target = a[k + 1]
                        break
                pc = labels[target]
            elif op == OP["Return"]:
                return None
            elif op == OP["ReturnValue"]:
                return val(a[0])
            elif op == OP["Kill"]:
                raise Discard()
            elif op == OP["Unreachable"]:
                raise RuntimeError("OpUnreachable executed")
            elif op == OP["Variable"]:
                self.m.vars.setdefault(a[1], (a[0], a[2]))
                self.store[a[1]] = m.zero(m.types[a[0]][1]) if len(a) < 4 else val(a[3])
                R[a[1]] = ("obj", a[1], ())
            elif op in (OP["AccessChain"], OP["InBoundsAccessChain"]):
                R[a[1]] = self._access(val(a[2]), [val(i) for i in a[3:]])
            elif op == OP["Load"]:
                R[a[1]] = self._load(val(a[2]))
            elif op == OP["Store"]:
                v = val(a[1])
                self._store(val(a[0]), list(v) if isinstance(v, list) else v)
            elif op == OP["ArrayLength"]:
                ptr = val(a[2])
                tid = m.types[ptr[3]][1][a[3]]
                off = m.mdecor[ptr[3]][a[3]][DEC["Offset"]][0]
                R[a[1]] = u32((len(self._buf(ptr[1])) - ptr[2] - off) // m.decor[tid][DEC["ArrayStride"]][0])
            elif op == OP["FunctionCall"]:
                R[a[1]] = self._call(a[2], [val(i) for i in a[3:]])
            elif op == OP["CompositeConstruct"]:
                t = m.types[a[0]]
                parts = [val(i) for i in a[2:]]
                if t[0] == "vector":
                    flat = []
                    for p in parts:
                        flat.extend(p if isinstance(p, list) else [p])
                    R[a[1]] = flat
                else:
                    R[a[1]] = parts
            elif op == OP["CompositeExtract"]:
                v = val(a[2])
                for i in a[3:]:
                    v = v[i]
                R[a[1]] = v
            elif op == OP["CompositeInsert"]:
                import copy
                comp = copy.deepcopy(val(a[3]))
                tgt = comp
                for i in a[4:-1]:
                    tgt = tgt[i]
                tgt[a[-1]] = val(a[2])
                R[a[1]] = comp
            elif op == OP["VectorShuffle"]:
                both = list(val(a[2])) + list(val(a[3]))
                R[a[1]] = [both[i] if i != 0xFFFFFFFF else f32(0) for i in a[4:]]
            elif op == OP["CopyObject"]:
                R[a[1]] = val(a[2])
            elif op == OP["FAdd"]:
                R[a[1]] = _map(lambda x, y: f32(x + y), val(a[2]), val(a[3]))
            elif op == OP["FSub"]:
                R[a[1]] = _map(lambda x, y: f32(x - y), val(a[2]), val(a[3]))
            elif op == OP["FMul"]:
                R[a[1]] = _map(lambda x, y: f32(x * y), val(a[2]), val(a[3]))
            elif op == OP["FDiv"]:
                R[a[1]] = _map(lambda x, y: f32(x / y), val(a[2]), val(a[3]))
            elif op == OP["FNegate"]:
                R[a[1]] = _map(lambda x: f32(-x), val(a[2]))
            elif op == OP["VectorTimesScalar"]:
                s = val(a[3])
                R[a[1]] = [f32(x * s) for x in val(a[2])]
            elif op == OP["MatrixTimesVector"]:
                M, v = val(a[2]), val(a[3])   # association order unspecified by SPIR-V: glam's, left to right
                rows = len(M[0])
                out = []
                for r in range(rows):
                    acc = f32(M[0][r] * v[0])
                    for c in range(1, len(M)):
                        acc = f32(f32(M[c][r] * v[c]) + acc)
                    out.append(acc)
                R[a[1]] = out
            elif op == OP["Dot"]:
                x, y = val(a[2]), val(a[3])
                acc = f32(x[0] * y[0])
                for k in range(1, len(x)):
                    acc = f32(acc + f32(x[k] * y[k]))
                R[a[1]] = acc
            elif op == OP["IAdd"]:
                R[a[1]] = _map(lambda x, y: type(x)((int(x) + int(y)) & 0xFFFFFFFF) if not isinstance(x, np.int32) else np.int32((int(x) + int(y) + 2**31) % 2**32 - 2**31), val(a[2]), val(a[3]))
            elif op == OP["ISub"]:
                R[a[1]] = _map(lambda x, y: u32((int(x) - int(y)) & 0xFFFFFFFF), val(a[2]), val(a[3]))
            elif op == OP["IMul"]:
                R[a[1]] = _map(lambda x, y: u32((int(x) * int(y)) & 0xFFFFFFFF), val(a[2]), val(a[3]))
            elif op == OP["UDiv"]:
                R[a[1]] = _map(lambda x, y: u32(int(x) // int(y)), val(a[2]), val(a[3]))
            elif op == OP["UMod"]:
                R[a[1]] = _map(lambda x, y: u32(int(x) % int(y)), val(a[2]), val(a[3]))
            elif op == OP["ShiftLeftLogical"]:
                R[a[1]] = _map(lambda x, y: type(x)((int(x) << int(y)) & 0xFFFFFFFF) if not isinstance(x, np.int32) else np.int32(((int(x) << int(y)) + 2**31) % 2**32 - 2**31), val(a[2]), val(a[3]))
            elif op == OP["ShiftRightLogical"]:
                R[a[1]] = _map(lambda x, y: u32((int(x) & 0xFFFFFFFF) >> int(y)), val(a[2]), val(a[3]))
            elif op == OP["BitwiseAnd"]:
                R[a[1]] = _map(lambda x, y: type(x)(int(x) & int(y)), val(a[2]), val(a[3]))
            elif op == OP["BitwiseOr"]:
                R[a[1]] = _map(lambda x, y: type(x)(int(x) | int(y)), val(a[2]), val(a[3]))
            elif op == OP["BitwiseXor"]:
                R[a[1]] = _map(lambda x, y: type(x)(int(x) ^ int(y)), val(a[2]), val(a[3]))
            elif op == OP["ConvertFToU"]:
                R[a[1]] = _map(_f_to_u, val(a[2]))
            elif op == OP["ConvertFToS"]:
                R[a[1]] = _map(lambda x: np.int32(max(min(int(float(x)) if np.isfinite(x) else 0, 2**31 - 1), -2**31)), val(a[2]))
            elif op == OP["ConvertUToF"]:
                R[a[1]] = _map(lambda x: f32(int(x) & 0xFFFFFFFF), val(a[2]))
            elif op == OP["ConvertSToF"]:
                R[a[1]] = _map(lambda x: f32(int(x)), val(a[2]))
            elif op == OP["Bitcast"]:
                src, t = val(a[2]), m.types[a[0]]

                def cast(x, t=t):
                    raw = struct.pack("<f", float(x)) if isinstance(x, np.floating) else struct.pack("<I", int(x) & 0xFFFFFFFF)
                    tt = m.types[t[1]] if t[0] == "vector" else t
                    if tt[0] == "float":
                        return f32(struct.unpack("<f", raw)[0])
                    return np.int32(struct.unpack("<i", raw)[0]) if tt[2] else u32(struct.unpack("<I", raw)[0])
                R[a[1]] = _map(cast, src)
            elif op == OP["Select"]:
                c, x, y = val(a[2]), val(a[3]), val(a[4])
                if isinstance(c, list):
                    R[a[1]] = [x[i] if c[i] else y[i] for i in range(len(c))]
                else:
                    R[a[1]] = x if c else y
            elif op == OP["LogicalOr"]:
                R[a[1]] = bool(val(a[2])) or bool(val(a[3]))
            elif op == OP["LogicalAnd"]:
                R[a[1]] = bool(val(a[2])) and bool(val(a[3]))
            elif op == OP["LogicalNot"]:
                R[a[1]] = not bool(val(a[2]))
            elif op == OP["LogicalEqual"]:
                R[a[1]] = bool(val(a[2])) == bool(val(a[3]))
            elif op == OP["LogicalNotEqual"]:
                R[a[1]] = bool(val(a[2])) != bool(val(a[3]))
            elif op in (OP["IEqual"], OP["INotEqual"]):
                eq = int(val(a[2])) & 0xFFFFFFFF == int(val(a[3])) & 0xFFFFFFFF
                R[a[1]] = eq if op == OP["IEqual"] else not eq
            elif op in (OP["ULessThan"], OP["ULessThanEqual"], OP["UGreaterThan"], OP["UGreaterThanEqual"]):
                x, y = int(val(a[2])) & 0xFFFFFFFF, int(val(a[3])) & 0xFFFFFFFF
                R[a[1]] = {OP["ULessThan"]: x < y, OP["ULessThanEqual"]: x <= y, OP["UGreaterThan"]: x > y,
                           OP["UGreaterThanEqual"]: x >= y}[op]
            elif op in (OP["SLessThan"], OP["SLessThanEqual"], OP["SGreaterThan"], OP["SGreaterThanEqual"]):
                x, y = int(np.int32(val(a[2]))), int(np.int32(val(a[3])))
                R[a[1]] = {OP["SLessThan"]: x < y, OP["SLessThanEqual"]: x <= y, OP["SGreaterThan"]: x > y,
                           OP["SGreaterThanEqual"]: x >= y}[op]
            elif OP["FOrdEqual"] <= op <= OP["FUnordGreaterThanEqual"]:
                x, y = float(val(a[2])), float(val(a[3]))
                unordered = x != x or y != y
                base = (op - OP["FOrdEqual"]) // 2
                is_unord = (op - OP["FOrdEqual"]) % 2 == 1
                res = [x == y, x != y, x < y, x > y, x <= y, x >= y][base]
                R[a[1]] = (True if is_unord else False) if unordered else res
            elif op == OP["IsNan"]:
                R[a[1]] = _map(lambda x: bool(np.isnan(x)), val(a[2]))
            elif op == OP["IsInf"]:
                R[a[1]] = _map(lambda x: bool(np.isinf(x)), val(a[2]))
            elif op == OP["ExtInst"]:
                R[a[1]] = self._ext(a[3], [val(i) for i in a[4:]])
            elif op == OP["SampledImage"]:
                R[a[1]] = ("sampled", val(a[2]), val(a[3]))
            elif op == OP["ImageSampleImplicitLod"]:
                si = val(a[2])
                R[a[1]] = [f32(x) for x in self.sample_cb("implicit", si[1][1:], si[2][1:], val(a[3]), None)]
            elif op == OP["ImageSampleExplicitLod"]:
                si = val(a[2])
                assert a[4] == 0x2, "only the Lod image operand is supported"
                R[a[1]] = [f32(x) for x in self.sample_cb("lod", si[1][1:], si[2][1:], val(a[3]), val(a[5]))]
            elif op in (OP["DPdx"], OP["DPdy"]):
                R[a[1]] = self.deriv_cb("dx" if op == OP["DPdx"] else "dy", val(a[2]))
            elif op in (OP["AtomicIIncrement"], OP["AtomicIAdd"]):
                ptr = val(a[2])
                old = self._load(ptr)
                inc = 1 if op == OP["AtomicIIncrement"] else int(val(a[5]))
                self._store(ptr, u32((int(old) + inc) & 0xFFFFFFFF))
                R[a[1]] = old
            else:
                name = {v: k for k, v in OP.items()}.get(op, op)
                raise NotImplementedError(f"opcode {name}")

    def _ext(self, inst, x):
        one = lambda fn: _map(lambda v: f32(fn(f32(v))), x[0])   # noqa: E731
        if inst == 31:
            return one(np.sqrt)
        if inst == 26:
            return _map(lambda b, e: f32(np.power(f32(b), f32(e))), x[0], x[1])
        if inst == 30:
            return one(np.log2)
        if inst == 28:
            return one(np.log)
        if inst == 27:
            return one(np.exp)
        if inst == 29:
            return one(np.exp2)
        if inst == 40:
            return _map(lambda p, q: f32(np.fmax(p, q)), x[0], x[1])
        if inst == 37:
            return _map(lambda p, q: f32(np.fmin(p, q)), x[0], x[1])
        if inst == 4:
            return one(np.abs)
        if inst == 13:
            return one(np.sin)
        if inst == 14:
            return one(np.cos)
        if inst == 15:
            return one(np.tan)
        if inst == 8:
            return one(np.floor)
        raise NotImplementedError(f"GLSL.std.450 {inst}")
