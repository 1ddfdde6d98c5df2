"""rust/cubecl-mi355 cannot be compiled in this image (no cargo / rustc).  What a compiler's name resolution would have
caught is checked here instead, against the reference checkout: every `cubecl_*::path::Item` the crate imports or spells
out is defined in that crate of /root/reference; every method it calls on the reference's memory pools, stream pool, drop
queue, capture state, metadata cache and profiler exists there with that many arguments; every `impl Trait for ...`
implements exactly the trait's items (all required ones, no invented ones, same parameter count); every field of a
reference struct / enum variant it builds exists; every `cubecl_hip_sys::` symbol is one the reference itself uses or one
hiprtc.h declares.  Skipped when the reference is not there (the GPU box)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SHIM = ROOT / "rust" / "cubecl-mi355" / "src"
REF = Path("/root/reference/crates")

pytestmark = pytest.mark.skipif(not REF.exists(), reason="reference checkout not present")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def shim_sources():
    return {p.name: strip_comments(p.read_text()) for p in sorted(SHIM.glob("*.rs")) if p.name != "ffi.rs"}


_crate_text = {}


def crate_text(crate):
    """All of a reference crate's source in one string (comments stripped)."""
    if crate not in _crate_text:
        root = REF / crate.replace("_", "-") / "src"
        assert root.exists(), f"the reference has no crate {crate}"
        _crate_text[crate] = "\n".join(strip_comments(p.read_text()) for p in sorted(root.rglob("*.rs")))
    return _crate_text[crate]


# crates that re-export wholesale from others: an item imported through them may be defined further down
REEXPORTS = {
    "cubecl_core": ["cubecl_runtime", "cubecl_ir", "cubecl_common", "cubecl_environment", "cubecl_zspace"],
    "cubecl_common": ["cubecl_environment"],
    "cubecl_runtime": [],
    "cubecl_ir": [],
    "cubecl_cpp": [],
    "cubecl_std": [],
    "cubecl_zspace": [],
    "cubecl_environment": [],
}


def defines(crate, item):
    pats = [rf"\b(?:pub(?:\([a-z]+\))?\s+)?(?:unsafe\s+)?(?:struct|enum|trait|type|fn|const|static|mod|union)\s+{re.escape(item)}\b",
            rf"macro_rules!\s+{re.escape(item)}\b",
            rf"pub use [^;]*\b{re.escape(item)}\b",
            rf"^\s*{re.escape(item)}\s*(?:\(|\{{|,|=)",       # enum variant
            rf"_type!\(\s*{re.escape(item)}\s*\)"]             # storage_id_type!(StorageId)
    for c in [crate] + REEXPORTS.get(crate, []):
        text = crate_text(c)
        if any(re.search(p, text, flags=re.M) for p in pats):
            return True
    return False


def expand_use(tree, prefix=()):
    """`a::{b, c::{d, e}, f}` -> [('a','b'), ('a','c','d'), ...] (handles `self`, `as`, `r#type`)."""
    tree = tree.strip()
    out = []
    depth = 0
    head = ""
    i = 0
    # split the top level on commas
    parts, cur = [], ""
    for ch in tree:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    if len(parts) > 1:
        for p in parts:
            out += expand_use(p, prefix)
        return out
    part = parts[0].strip() if parts else ""
    if not part:
        return []
    m = re.match(r"^([^{]*?)::\{(.*)\}$", part, flags=re.S)
    if m:
        segs = tuple(s.strip() for s in m.group(1).split("::") if s.strip())
        return expand_use(m.group(2), prefix + segs)
    part = re.sub(r"\s+as\s+\w+$", "", part)
    segs = tuple(s.strip().replace("r#", "") for s in part.split("::") if s.strip())
    if segs and segs[-1] == "self":
        segs = segs[:-1]
    return [prefix + segs]


def test_every_imported_or_spelled_out_reference_path_exists():
    missing = []
    seen = 0
    for name, text in shim_sources().items():
        paths = []
        for m in re.finditer(r"\buse\s+((?:cubecl_\w+)[^;]*);", text, flags=re.S):
            paths += expand_use(m.group(1))
        for m in re.finditer(r"\b(cubecl_\w+(?:::(?:r#)?\w+)+)", re.sub(r"\buse\s+[^;]*;", "", text, flags=re.S)):
            paths.append(tuple(s.replace("r#", "") for s in m.group(1).split("::")))
        for path in paths:
            crate, rest = path[0], path[1:]
            if crate == "cubecl_hip_sys" or not rest:
                continue
            seen += 1
            # every segment after the crate has to be something that crate (or what it re-exports) defines
            for seg in rest:
                if seg in ("prelude",):
                    continue
                if not defines(crate, seg):
                    missing.append(f"{name}: {'::'.join(path)} ({seg})")
    assert seen > 120, seen
    assert not missing, "\n".join(missing)


def find_fns(text):
    """(name, start, args, end-of-parameter-list) for every `fn` in `text`; generics may contain `Fn() -> T`."""
    for m in re.finditer(r"\bfn\s+(\w+)\s*", text):
        i = m.end()
        if i < len(text) and text[i] == "<":
            depth = 0
            while i < len(text):
                ch = text[i]
                if ch == "<":
                    depth += 1
                elif ch == ">" and text[i - 1] != "-":
                    depth -= 1
                    if depth == 0:
                        i += 1
                        break
                i += 1
            while i < len(text) and text[i].isspace():
                i += 1
        if i >= len(text) or text[i] != "(":
            continue
        args = balanced(text, i)
        yield m.group(1), m.start(), args, i + len(args) + 2


def fn_table(path, type_hint=None):
    """{name: [param counts]} of the `fn`s in a reference file (self not counted)."""
    text = strip_comments(Path(path).read_text())
    table = {}
    for name, _, args, _ in find_fns(text):
        table.setdefault(name, []).append(count_params(args))
    return table


def balanced(text, open_idx):
    depth = 0
    for i in range(open_idx, len(text)):
        if text[i] in "([{":
            depth += 1
        elif text[i] in ")]}":
            depth -= 1
            if depth == 0:
                return text[open_idx + 1:i]
    raise AssertionError("unbalanced")


def split_top(args):
    parts, cur, depth = [], "", 0
    for i, ch in enumerate(args):
        if ch in "([{<":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == ">" and not (i and args[i - 1] in "-="):
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts if p.strip()]


def count_params(args):
    parts = split_top(args)
    return len([p for p in parts if not re.match(r"^(?:&\s*(?:'\w+\s+)?)?(?:mut\s+)?self\b", p)])


RT = REF / "cubecl-runtime" / "src"
RECEIVERS = {
    # receiver spelling in the crate -> reference file whose `pub fn`s it may call
    r"\.memory_management_gpu": RT / "memory_management" / "memory_manage.rs",
    r"\.memory_management_cpu": RT / "memory_management" / "memory_manage.rs",
    r"\.drop_queue": RT / "memory_management" / "drop_queue" / "queue.rs",
    r"\.capturing": RT / "stream" / "capture.rs",
    r"\.info_cache": RT / "metadata_cache.rs",
    r"\bself\.timestamps": RT / "timestamp_profiler.rs",
    r"\bself\.lanes": RT / "stream" / "event.rs",
    r"\bpass\.lanes": RT / "stream" / "event.rs",
    r"\blanes": RT / "stream" / "event.rs",
}


def test_methods_called_on_reference_objects_exist_with_that_arity():
    problems, checked = [], 0
    for name, text in shim_sources().items():
        for recv, ref_file in RECEIVERS.items():
            table = fn_table(ref_file)
            for m in re.finditer(recv + r"\s*\.\s*(\w+)\s*\(", text):
                method = m.group(1)
                if recv.endswith("lanes") and method in ("current", "get", "cursor"):
                    pass
                args = balanced(text, m.end() - 1)
                n = len(split_top(args))
                checked += 1
                if method not in table:
                    problems.append(f"{name}: {recv}.{method}() is not a fn of {ref_file.name}")
                elif n not in table[method]:
                    problems.append(f"{name}: {recv}.{method}() called with {n} argument(s), {ref_file.name} takes {table[method]}")
    assert checked >= 45, checked
    assert not problems, "\n".join(problems)


def test_public_fields_read_from_reference_objects_exist():
    """`.lanes.cursor`, `.lanes.current` (fields of ResolvedStreams), `.logger` of MultiStream, fields of Handle / BufferBinding."""
    event = strip_comments((RT / "stream" / "event.rs").read_text())
    resolved = re.search(r"pub struct ResolvedStreams<[^{]*\{(.*?)\n\}", event, flags=re.S).group(1)
    multi = re.search(r"pub struct MultiStream<[^{]*\{(.*?)\n\}", event, flags=re.S).group(1)
    assert re.search(r"pub cursor\s*:", resolved) and re.search(r"pub current\s*:", resolved)
    assert re.search(r"pub logger\s*:", multi)
    handle = strip_comments((RT / "server" / "handle.rs").read_text())
    for field in ("memory", "offset_start", "offset_end", "stream"):
        assert re.search(rf"pub {field}\s*:", handle), field


TRAITS = {
    # trait name -> file that defines it
    "ComputeServer": RT / "server" / "base.rs",
    "ServerCommunication": RT / "server" / "base.rs",
    "ComputeStorage": RT / "storage" / "base.rs",
    "EventStreamBackend": RT / "stream" / "event.rs",
    "Compiler": RT / "compiler.rs",
    "CubeTask": RT / "compiler.rs",
    "KernelMetadata": RT / "kernel.rs",
    "Runtime": RT / "runtime.rs",
    "DeviceService": REF / "cubecl-common" / "src" / "device" / "base.rs",
    "Device": REF / "cubecl-common" / "src" / "device" / "base.rs",
    "AllocationController": REF / "cubecl-environment" / "src" / "bytes" / "base.rs",
    "Fence": RT / "memory_management" / "drop_queue" / "queue.rs",
}


def trait_items(trait, path):
    text = strip_comments(Path(path).read_text())
    m = re.search(rf"pub trait {trait}\b[^{{]*\{{", text)
    assert m, f"{trait} not in {path}"
    body = balanced(text, m.end() - 1)
    items = {}
    for fname, _, args, end in find_fns(body):
        rest = body[end:]
        # a required method ends in `;` before any `{`
        semi, brace = rest.find(";"), rest.find("{")
        required = semi != -1 and (brace == -1 or semi < brace)
        items[fname] = (count_params(args), required)
    types = {t: "=" not in rest for t, rest in re.findall(r"\btype\s+(\w+)([^;]*);", body)}
    consts = re.findall(r"\bconst\s+(\w+)\s*:", body)
    return items, types, consts


def test_trait_impls_match_the_reference_traits():
    problems, impls = [], 0
    for name, text in shim_sources().items():
        for m in re.finditer(r"\bimpl(?:<[^>]*>)?\s+(?:[\w:]+::)?(\w+)(?:<[^>{]*>)?\s+for\s+([\w<>:, ]+?)\s*\{", text):
            trait = m.group(1)
            if trait not in TRAITS:
                continue
            impls += 1
            body = balanced(text, m.end() - 1)
            want_fns, want_types, want_consts = trait_items(trait, TRAITS[trait])
            have = {}
            for fname, start, args, _ in find_fns(body):
                # only the impl's own items, not fns nested in bodies: those start at brace depth 0 of `body`
                if body[:start].count("{") != body[:start].count("}"):
                    continue
                have[fname] = count_params(args)
            for fn, n in have.items():
                if fn not in want_fns:
                    problems.append(f"{name}: impl {trait} for {m.group(2)} defines {fn}(), which {trait} does not have")
                elif want_fns[fn][0] != n:
                    problems.append(f"{name}: {trait}::{fn} takes {want_fns[fn][0]} parameter(s) in the reference, {n} here")
            for fn, (_, required) in want_fns.items():
                if required and fn not in have:
                    problems.append(f"{name}: impl {trait} for {m.group(2)} lacks the required {fn}()")
            have_types = set(re.findall(r"^\s*type\s+(\w+)\s*=", body, flags=re.M))
            for t, required in want_types.items():
                if required and t not in have_types:
                    problems.append(f"{name}: impl {trait} for {m.group(2)} lacks `type {t}`")
            for t in have_types - set(want_types):
                problems.append(f"{name}: impl {trait} for {m.group(2)} defines `type {t}`, unknown to {trait}")
            for c in want_consts:
                if not re.search(rf"\bconst\s+{c}\s*:", body):
                    problems.append(f"{name}: impl {trait} for {m.group(2)} lacks `const {c}`")
    assert impls >= 13, impls
    assert not problems, "\n".join(problems)


def _norm_type(ty):
    ty = re.sub(r"'\w+\s*", "", ty)                    # lifetimes
    ty = re.sub(r"\s+", "", ty)
    ty = ty.replace("crate::", "").replace("self::", "")
    return re.sub(r"(\w+::)+(\w+)", r"\2", ty)          # module paths: the last segment names the type


def _signature(text, args, end):
    params = []
    for a in split_top(args):
        a = a.strip()
        if not a:
            continue
        if re.match(r"^(&\s*('\w+\s*)?)?(mut\s+)?self\b", a):
            params.append(re.sub(r"\s+", "", re.sub(r"'\w+\s*", "", a)))
        else:
            params.append(_norm_type(a.partition(":")[2]))
    m = re.match(r"\s*->\s*([^{;]+?)\s*(where\b[^{;]*)?[{;]", text[end:], flags=re.S)
    return params, (_norm_type(m.group(1)) if m else "")


# the only places where an impl may spell a type differently from the trait: an associated or generic type written out
SIGNATURE_SUBSTITUTIONS = {
    ("ComputeServer", "get_resource"): ("<StorageasComputeStorage>::Resource", "DeviceSlice"),        # type Storage = DeviceStorage
    ("CubeTask", "compile"): None,                                                                      # C = Mi355Compiler, see below
}


def test_trait_impl_signatures_are_the_reference_signatures_type_for_type():
    """Parameter types and return types of every method the crate implements for a reference trait, compared as text with
    the trait's own declaration (whitespace, lifetimes and module paths aside).  What a first `cargo check` would say about
    the method headers, as far as text can."""
    problems, compared = [], 0
    for name, text in shim_sources().items():
        for m in re.finditer(r"\bimpl(?:<[^>]*>)?\s+(?:[\w:]+::)?(\w+)(?:<[^>{]*>)?\s+for\s+([\w<>:, ]+?)\s*\{", text):
            trait = m.group(1)
            if trait not in TRAITS:
                continue
            body = balanced(text, m.end() - 1)
            ttext = strip_comments(Path(TRAITS[trait]).read_text())
            tm = re.search(rf"pub trait {trait}\b[^{{]*\{{", ttext)
            tbody = balanced(ttext, tm.end() - 1)
            want = {f: _signature(tbody, a, e) for f, st, a, e in find_fns(tbody) if tbody[:st].count("{") == tbody[:st].count("}")}
            for fname, start, args, end in find_fns(body):
                if body[:start].count("{") != body[:start].count("}") or fname not in want:
                    continue
                have, ref = _signature(body, args, end), want[fname]
                compared += 1
                if have == ref:
                    continue
                sub = SIGNATURE_SUBSTITUTIONS.get((trait, fname), ())
                if sub is None:      # CubeTask<C>::compile with C = Mi355Compiler: `&mut C`, `&C::CompilationOptions`, `CompiledKernel<C>`
                    ok = have == (["&self", "KernelDefinition", "&mutMi355Compiler", "&<Mi355CompilerasCompiler>::CompilationOptions"],
                                  "Result<CompiledKernel<Mi355Compiler>,CompilationError>") and \
                         ref == (["&self", "KernelDefinition", "&mutC", "&CompilationOptions"], "Result<CompiledKernel<C>,CompilationError>")
                elif sub:
                    ok = ([p.replace(sub[0], sub[1]) for p in ref[0]], ref[1].replace(sub[0], sub[1])) == have
                else:
                    ok = False
                if not ok:
                    problems.append(f"{name}: {trait}::{fname}\n    reference: ({', '.join(ref[0])}) -> {ref[1]}\n    crate    : ({', '.join(have[0])}) -> {have[1]}")
    assert compared >= 60, compared
    assert not problems, "\n".join(problems)


# struct / enum-variant literals of reference types the crate builds: name -> (file, kind)
LITERALS = {
    "CompiledKernel": RT / "kernel.rs",
    "KernelDefinition": RT / "kernel.rs",
    "HardwareProperties": REF / "cubecl-ir" / "src" / "properties.rs",
    "MemoryDeviceProperties": REF / "cubecl-ir" / "src" / "properties.rs",
    "DeviceIdentity": REF / "cubecl-ir" / "src" / "properties.rs",
    "MmaProperties": REF / "cubecl-ir" / "src" / "runtime_properties.rs",
    "TargetProperties": REF / "cubecl-ir" / "src" / "runtime_properties.rs",
    "MmaConfig": REF / "cubecl-ir" / "src" / "features.rs",
    "ScaledMmaConfig": REF / "cubecl-ir" / "src" / "features.rs",
    "CompilationOptions": REF / "cubecl-cpp" / "src" / "shared" / "base.rs",
    "StorageUtilization": RT / "storage" / "base.rs",
    "DeviceId": REF / "cubecl-common" / "src" / "device" / "base.rs",
    "CopyDescriptor": RT / "server" / "base.rs",
    "KernelArguments": RT / "server" / "base.rs",
}
VARIANTS = {
    # Enum::Variant { fields } -> file
    "ServerError": RT / "server" / "base.rs",
    "IoError": RT / "server" / "base.rs",
    "LaunchError": RT / "server" / "base.rs",
    "ResourceLimitError": RT / "server" / "base.rs",
    "ProfileError": RT / "server" / "base.rs",
    "CompilationError": RT / "compiler.rs",
}


def struct_fields(path, name):
    text = strip_comments(Path(path).read_text())
    m = re.search(rf"pub struct {name}\b[^{{;]*\{{", text)
    assert m, f"struct {name} not in {path}"
    return set(re.findall(r"(?:pub(?:\([a-z]+\))?\s+)?(\w+)\s*:", re.sub(r"#\[[^\]]*\]", "", balanced(text, m.end() - 1))))


def variant_fields(path, enum, variant):
    text = strip_comments(Path(path).read_text())
    m = re.search(rf"pub enum {enum}\b[^{{]*\{{", text)
    assert m, f"enum {enum} not in {path}"
    body = re.sub(r"#\[(?:[^\[\]]|\[[^\]]*\])*\]", "", balanced(text, m.end() - 1))
    v = re.search(rf"\b{variant}\s*(\{{|\(|,|\n)", body)
    if not v:
        return None
    if v.group(1) != "{":
        return set()
    return set(re.findall(r"(\w+)\s*:", balanced(body, v.end() - 1)))


def literal_field_names(body):
    names = []
    for part in split_top(body):
        if part.startswith(".."):
            continue
        m = re.match(r"^(\w+)\s*(?::|$)", part)
        if m:
            names.append(m.group(1))
    return names


def test_fields_of_reference_structs_and_error_variants_exist():
    problems, checked = [], 0
    for name, text in shim_sources().items():
        for struct, path in LITERALS.items():
            for m in re.finditer(rf"(?<![\w:]){struct}\s*\{{", text):
                before = text[max(0, m.start() - 12):m.start()]
                if re.search(r"(struct|enum|impl|for|->)\s*$", before):
                    continue
                fields = struct_fields(path, struct)
                for f in literal_field_names(balanced(text, m.end() - 1)):
                    checked += 1
                    if f not in fields:
                        problems.append(f"{name}: {struct} has no field `{f}` in the reference")
        for enum, path in VARIANTS.items():
            for m in re.finditer(rf"\b{enum}::(\w+)\s*\{{", text):
                fields = variant_fields(path, enum, m.group(1))
                if fields is None:
                    problems.append(f"{name}: {enum}::{m.group(1)} is not a variant in the reference")
                    continue
                for f in literal_field_names(balanced(text, m.end() - 1)):
                    checked += 1
                    if f not in fields:
                        problems.append(f"{name}: {enum}::{m.group(1)} has no field `{f}` in the reference")
            for m in re.finditer(rf"\b{enum}::(\w+)\b(?!\s*\{{)", text):
                if variant_fields(path, enum, m.group(1)) is None and not re.search(rf"fn {m.group(1)}\b", strip_comments(Path(path).read_text())):
                    problems.append(f"{name}: {enum}::{m.group(1)} is neither a variant nor an associated fn in the reference")
    assert checked >= 90, checked
    assert not problems, "\n".join(problems)


def test_free_functions_and_constructors_are_called_with_the_reference_arity():
    """`Type::function(args)` for the reference types the crate constructs or calls statically."""
    calls = {
        # spelled in the crate : (file, fn name)
        "MultiStream::new": (RT / "stream" / "event.rs", "new", 3),
        "MemoryManagement::from_configuration": (RT / "memory_management" / "memory_manage.rs", "from_configuration", 5),
        "MemoryManagementOptions::new": (RT / "memory_management" / "memory_manage.rs", "new", 1),
        "PendingDropQueue::new": (RT / "memory_management" / "drop_queue" / "queue.rs", "new", 1),
        "MetadataInfoCache::new": (RT / "metadata_cache.rs", "new", 1),
        "ServerUtilities::new": (RT / "server" / "base.rs", "new", 4),
        "PitchedMemoryLayoutPolicy::new": (RT / "allocator.rs", "new", 1),
        "DeviceProperties::new": (REF / "cubecl-ir" / "src" / "properties.rs", "new", 5),
        "Handle::new": (RT / "server" / "handle.rs", "new", 2),
        "ManagedResource::new": (RT / "storage" / "base.rs", None, 2),        # #[derive(new)]: binding, resource
        "StorageHandle::new": (RT / "storage" / "base.rs", None, 2),          # #[derive(new)]: id, utilization
        "CopyDescriptor::new": (RT / "server" / "base.rs", None, 4),          # #[derive(new)]: handle, shape, strides, elem_size
        "KernelSettings::new": (REF / "cubecl-ir" / "src" / "settings.rs", "new", 3),
        "Scope::root": (REF / "cubecl-ir" / "src" / "scope.rs", "root", 1),
        "ContiguousElements::new": (REF / "cubecl-ir" / "src" / "runtime_properties.rs", "new", 1),
        "ComputeClient::load": (RT / "client.rs", "load", 1),
        "Bytes::from_controller": (REF / "cubecl-environment" / "src" / "bytes" / "base.rs", "from_controller", 2),
        "Bytes::from_bytes_vec": (REF / "cubecl-environment" / "src" / "bytes" / "base.rs", "from_bytes_vec", 1),
        "ServerError::graph_state": (RT / "server" / "base.rs", "graph_state", 1),
        "validate_cube_dim": (RT / "validation.rs", "validate_cube_dim", 2),
        "validate_units": (RT / "validation.rs", "validate_units", 2),
        "has_pitched_row_major_strides": (REF / "cubecl-zspace" / "src" / "striding" / "layout_validation.rs", "has_pitched_row_major_strides", 2),
        "matrix_batch_layout": (REF / "cubecl-std" / "src" / "tensor" / "matrix_batch_layout.rs", "matrix_batch_layout", 2),
        "Type::atomic": (REF / "cubecl-ir" / "src" / "type.rs", "atomic", 1),
    }
    all_text = "\n".join(shim_sources().values())
    problems, used = [], 0
    for spelled, (path, fn, arity) in calls.items():
        sites = list(re.finditer(rf"(?<![\w:]){re.escape(spelled)}\s*\(", all_text))
        if not sites:
            continue
        used += 1
        if fn is not None:
            table = fn_table(path)
            if fn not in table or arity not in table[fn]:
                problems.append(f"{spelled}: the reference's {path.name} has {table.get(fn)} parameter counts for `{fn}`, expected {arity}")
        else:
            struct = spelled.split("::")[0]
            text = strip_comments(Path(path).read_text())
            m = re.search(rf"#\[derive\([^)]*\bnew\b[^)]*\)\]\s*pub struct {struct}\b[^{{]*\{{", text)
            if not m:
                problems.append(f"{struct} does not derive `new` in the reference")
            elif len(split_top(re.sub(r"#\[[^\]]*\]", "", balanced(text, m.end() - 1)))) != arity:
                problems.append(f"{struct}::new arity differs from {arity}")
        for site in sites:
            n = len(split_top(balanced(all_text, site.end() - 1)))
            if n != arity:
                problems.append(f"{spelled} called with {n} argument(s), the reference takes {arity}")
    assert used >= 20, used
    assert not problems, "\n".join(problems)


def test_client_methods_the_launchers_use_exist():
    client = fn_table(RT / "client.rs")
    text = shim_sources()["ops.rs"]
    for m in re.finditer(r"\bclient\.(\w+)\s*\(", text):
        n = len(split_top(balanced(text, m.end() - 1)))
        assert m.group(1) in client and n in client[m.group(1)], (m.group(1), n, client.get(m.group(1)))


def test_hip_sys_symbols_are_known():
    used = set()
    for text in shim_sources().values():
        for m in re.finditer(r"\buse\s+cubecl_hip_sys::\{([^}]*)\}", text):
            used |= {s.strip() for s in m.group(1).split(",") if s.strip()}
        used |= set(re.findall(r"\bcubecl_hip_sys::(\w+)", text))
    assert used, "program.rs is expected to use hiprtc through cubecl_hip_sys"
    ref_uses = "\n".join(p.read_text() for p in (REF / "cubecl-hip" / "src").rglob("*.rs"))
    hiprtc = Path("/opt/rocm/include/hip/hiprtc.h")
    header = hiprtc.read_text() if hiprtc.exists() else ""
    unknown = [s for s in sorted(used) if not re.search(rf"\b{s}\b", ref_uses) and not re.search(rf"\b{s}\b", header)
               and not (s.startswith("hiprtcResult_") and s[len("hiprtcResult_"):] in header)]
    assert not unknown, unknown


def test_conformance_suites_are_instantiated_like_the_reference_backends_do():
    lib = (SHIM / "lib.rs").read_text()
    ref = (REF / "cubecl-hip" / "src" / "lib.rs").read_text()
    for macro in ("cubecl_std::testgen!", "cubecl_core::testgen_all!", "cubecl_core::testgen_launch_dynamic_count!"):
        assert macro in ref and macro in lib, macro
    assert "pub type TestRuntime = crate::Mi355Runtime;" in lib
    # the macros exist under those names
    assert re.search(r"macro_rules!\s+testgen_all\b", crate_text("cubecl_core"))
    assert re.search(r"macro_rules!\s+testgen_launch_dynamic_count\b", crate_text("cubecl_core"))
    assert re.search(r"macro_rules!\s+testgen\b", crate_text("cubecl_std"))


def test_nothing_is_launched_on_a_null_stream_and_stream_ids_are_honoured():
    """Round-1 finding: every call passed `core::ptr::null_mut()` as the stream and ignored `stream_id`."""
    server = shim_sources()["server.rs"] + shim_sources()["comm.rs"]
    body = server[server.index("impl ComputeServer for Mi355Server"):]
    for m in re.finditer(r"\b(mi355_(?:gemm\w*|reduce\w*|argmax\w*|sum_argmax\w*|launch|write\w*|read\w*|graph_replay|all_reduce|send|recv|sync_collective))\s*\(", server):
        args = split_top(balanced(server, m.end() - 1))
        assert len(args) >= 2 and "null" not in args[1] and (len(args) < 3 or "null_mut" not in args[2] or m.group(1) in ("mi355_gemm",)), (m.group(1), args[:3])
    # every trait method that receives a stream_id resolves it
    for m in re.finditer(r"\bfn\s+(\w+)\s*\(([^)]*stream_id: StreamId[^)]*)\)[^{]*\{", body):
        fn_body = balanced(body, m.end() - 1)
        assert re.search(r"stream_id", fn_body), f"{m.group(1)} ignores its stream_id"


def test_integration_doc_lists_no_invented_symbols():
    doc = (ROOT / "INTEGRATION.md").read_text()
    for ghost in ("reserve_into", "keep_alive_until_flush", "upload_info", "with_server"):
        assert ghost not in doc, ghost
        for name, text in shim_sources().items():
            assert ghost not in text, (ghost, name)


# ---- round 3: the likeliest first compile errors a text check can still reach ---------------------------------------------
def _generic_arity(decl_generics):
    """(required, total) type / const parameters of a `<...>` declaration; lifetimes do not count, defaulted ones are optional."""
    params = [p for p in split_top(decl_generics) if not p.startswith("'")]
    required = len([p for p in params if "=" not in p.split(":")[0] and not re.search(r"=\s*[\w:<>]+\s*$", p)])
    return required, len(params)


def test_generic_types_are_spelled_with_the_reference_number_of_parameters():
    """`MemoryManagement<S>`, `MultiStream<B>`, `ComputeClient<R>`, `PendingDropQueue<F>` ...: every reference-defined generic
    type the crate spells with `<...>` carries as many type arguments as its declaration takes (lifetimes aside)."""
    all_ref = "\n".join(crate_text(c) for c in REEXPORTS)
    decls = {}
    for m in re.finditer(r"\bpub(?:\([a-z]+\))?\s+(?:struct|enum|trait|type)\s+(\w+)\s*<", all_ref):
        i = m.end() - 1
        depth, j = 0, i
        while j < len(all_ref):
            if all_ref[j] == "<":
                depth += 1
            elif all_ref[j] == ">" and all_ref[j - 1] != "-":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        decls.setdefault(m.group(1), set()).add(_generic_arity(all_ref[i + 1:j]))
    checked, problems = 0, []
    own = set(re.findall(r"\b(?:struct|enum|trait|type)\s+(\w+)", "\n".join(shim_sources().values())))
    own |= {"Result", "Option", "Vec", "Box", "Arc", "Rc", "HashMap", "HashSet", "BTreeMap", "VecDeque", "PhantomData", "Mutex", "RefCell", "Cell",
            "Cow", "Pin", "Future", "Iterator", "IntoIterator", "Fn", "FnMut", "FnOnce", "From", "Into", "AsRef", "Deref", "MaybeUninit"}   # std's, not the reference's
    for name, text in shim_sources().items():
        for m in re.finditer(r"\b([A-Z]\w+)\s*<", text):
            ident = m.group(1)
            if ident not in decls or ident in own or text[max(0, m.start() - 2):m.start()] == "::" and False:
                continue
            i = m.end() - 1
            depth, j = 0, i
            while j < len(text):
                if text[j] == "<":
                    depth += 1
                elif text[j] == ">" and text[j - 1] not in "-=":
                    depth -= 1
                    if depth == 0:
                        break
                elif text[j] in ";{" and depth == 1 and j - i > 200:
                    break
                j += 1
            if depth != 0:
                continue                                   # a comparison, not a generic argument list
            args = [a for a in split_top(text[i + 1:j]) if not a.startswith("'")]
            if any(re.match(r"^\w+\s*=", a) for a in args):        # associated-type bindings (`Iterator<Item = T>`)
                continue
            checked += 1
            if not any(req <= len(args) <= tot for req, tot in decls[ident]):
                problems.append(f"{name}: {ident}<{text[i + 1:j].strip()[:60]}> has {len(args)} type argument(s), the reference declares {sorted(decls[ident])}")
    assert checked >= 25, checked
    assert not problems, "\n".join(problems)


def test_every_mi355_constant_the_crate_uses_is_defined_in_ffi_rs_with_the_headers_value():
    """`MI355_ABI_VERSION` was used by runtime.rs and defined nowhere until round 3: a guaranteed first compile error."""
    ffi = strip_comments((SHIM / "ffi.rs").read_text())
    defined = dict(re.findall(r"pub const (MI355_\w+)\s*:\s*\w+\s*=\s*([^;]+);", ffi))
    used = set()
    for text in shim_sources().values():
        used |= set(re.findall(r"\b(MI355_[A-Z0-9_]+)\b", text))
    missing = sorted(u for u in used if u not in defined)
    assert not missing, f"used in src/*.rs but not defined in ffi.rs: {missing}"
    header = (ROOT / "include" / "mi355cube.h").read_text()
    hvals = dict(re.findall(r"#define\s+(MI355_\w+)\s+(-?\d+)\b", header))
    hvals.update(dict(re.findall(r"\b(MI355_\w+)\s*=\s*(-?\d+)\s*[,}/]", header)))
    wrong = {k: (v.strip(), hvals[k]) for k, v in defined.items() if k in hvals and re.fullmatch(r"-?\d+", v.strip()) and int(v) != int(hvals[k])}
    assert not wrong, wrong
    assert int(defined["MI355_ABI_VERSION"]) == int(hvals["MI355_ABI_VERSION"])


def _error_type(ret):
    m = re.search(r"Result<(.*)>\s*$", ret.strip(), flags=re.S)
    if not m:
        return None
    parts = split_top(m.group(1))
    return re.sub(r"\s+", "", parts[-1]).split("::")[-1] if len(parts) == 2 else None


def test_question_mark_sites_propagate_an_error_type_the_function_can_return():
    """`foo()?` inside `fn bar() -> Result<_, E>` compiles only when foo's error type is E or converts into it.  The reference
    gives ServerError `#[from]` conversions out of IoError / LaunchError / ProfileError -- not the other way round -- so the
    likeliest first compile error of this crate is an `IoError`-returning function using `?` on a ServerError.  Callee error
    types come from this crate's own signatures and from the reference's `fn` declarations (skipped when a name has
    several declarations with different error types, or when the `?` follows a closure-typed combinator)."""
    rt = crate_text("cubecl_runtime") + "\n" + crate_text("cubecl_common")
    conversions = set()
    for enum_m in re.finditer(r"pub enum (\w+Error)\b[^{]*\{", rt):
        body = balanced(rt, enum_m.end() - 1)
        for boxed, frm in re.findall(r"#\[from\]\s*(Box<)?(\w+)", body):
            conversions.add((f"Box<{frm}>" if boxed else frm, enum_m.group(1)))      # From<Box<E>> is not From<E>
    conversions |= set(re.findall(r"impl From<(\w+)> for (\w+)", rt))
    assert ("IoError", "ServerError") in conversions and ("ServerError", "IoError") not in conversions

    def declared_errors(text):
        out = {}
        for fname, start, args, end in find_fns(text):
            m = re.match(r"\s*->\s*([^{;]+?)\s*(?:where\b[^{;]*)?[{;]", text[end:], flags=re.S)
            e = _error_type(m.group(1)) if m else None
            if e:
                out.setdefault(fname, set()).add(e)
        return out
    own, ref = {}, declared_errors(rt)
    sources = shim_sources()
    for text in sources.values():
        for k, v in declared_errors(text).items():
            own.setdefault(k, set()).update(v)
    COMBINATORS = {"ok_or_else", "ok_or", "map_err", "map", "and_then", "transpose", "collect", "unwrap_or", "into", "ok", "get", "get_mut", "next"}
    checked, problems = 0, []
    for name, text in sources.items():
        for fname, start, args, end in find_fns(text):
            m = re.match(r"\s*->\s*([^{;]+?)\s*(?:where\b[^{;]*)?\{", text[end:], flags=re.S)
            want = _error_type(m.group(1)) if m else None
            if not want:
                continue
            body_open = end + m.end() - 1
            body = balanced(text, body_open)
            for q in re.finditer(r"\)\s*\?", body):
                # the call whose result the `?` applies to: walk back over the balanced argument list to its name
                j, depth = q.start(), 0
                while j >= 0:
                    if body[j] == ")":
                        depth += 1
                    elif body[j] == "(":
                        depth -= 1
                        if depth == 0:
                            break
                    j -= 1
                cm = re.search(r"(\w+)\s*(?:::<[^>]*>)?\s*$", body[:j])
                if not cm or cm.group(1) in COMBINATORS:
                    continue
                callee = cm.group(1)
                errs = own.get(callee) or ref.get(callee)
                if not errs or len(errs) != 1:
                    continue
                have = next(iter(errs))
                checked += 1
                if have != want and (have, want) not in conversions:
                    problems.append(f"{name}: fn {fname} -> Result<_, {want}> uses `?` on {callee}(..) which returns Result<_, {have}> (no From<{have}> for {want})")
    assert checked >= 25, checked
    assert not problems, "\n".join(problems)
