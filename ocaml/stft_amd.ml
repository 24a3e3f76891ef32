(* stft_amd.ml -- the OCaml side of the drop-in: the bodies of stft.ml / mel.ml / soundml.ml / convert.ml /
   spectral.ml / chroma.ml / resample.ml that the hot path funnels through, re-expressed over the C ABI of
   include/soundml_amd.h (stubs: soundml_amd_stubs.c).  Everything else in Soundml (Config validation, the frame
   grid, Pipeline, the resample planner) stays as it is; every public signature is unchanged.

   NOT compiled in this repository (no OCaml toolchain in the build image); INTEGRATION.md lists, entry for entry,
   which definition of the reference each one replaces, and tests/test_host_logic.py checks that every C entry point
   named here and in the stubs exists in the header with the arity used. *)

type stft_handle
type mel_handle
type chroma_handle
type kernel_handle
type fir_handle
type stage_handle

type ('a, 'b) flat = ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t

(* ---- externals (native name second; the bytecode shim is needed above five arguments, resample.ml:1162-1196) ---- *)

external stft_config_c :
  int -> int -> int -> int -> int -> float -> int -> (float, Bigarray.float64_elt) flat -> stft_handle
  = "soundml_amd_stft_config_bc" "soundml_amd_stft_config"

(* x, out: flat views of contiguous nx storage; lead, n, p0, p1, mode (0 complex, 1 |.|^power), power *)
external stft_range_c :
  stft_handle -> ('a, 'b) flat -> ('c, 'd) flat -> int -> int -> int -> int -> int -> float -> unit
  = "soundml_amd_stft_range_bc" "soundml_amd_stft_range"

external stft_invert_c : stft_handle -> ('a, 'b) flat -> ('c, 'd) flat -> int -> int -> int -> int -> unit
  = "soundml_amd_stft_invert_bc" "soundml_amd_stft_invert"

(* s, init (empty = default phase), out; lead, bins, frames, n_iter, momentum, length (-1 = not given) *)
external griffin_lim_c :
  stft_handle -> ('a, 'b) flat -> ('a, 'b) flat -> ('a, 'b) flat -> int -> int -> int -> int -> float -> int -> unit
  = "soundml_amd_griffin_lim_bc" "soundml_amd_griffin_lim"

external mel_config_c : int -> int -> int -> float -> float -> int -> int -> mel_handle
  = "soundml_amd_mel_config_bc" "soundml_amd_mel_config"

external mel_apply_c : mel_handle -> ('a, 'b) flat -> ('a, 'b) flat -> int -> int -> int -> unit
  = "soundml_amd_mel_apply_bc" "soundml_amd_mel_apply"

external mel_spectrogram_c :
  stft_handle -> mel_handle -> ('a, 'b) flat -> ('a, 'b) flat -> int -> int -> float -> unit
  = "soundml_amd_mel_spectrogram_bc" "soundml_amd_mel_spectrogram"

(* x, out; lead, n, n_mfcc, lifter (-1. = not given) *)
external mfcc_c : stft_handle -> mel_handle -> ('a, 'b) flat -> ('a, 'b) flat -> int -> int -> int -> float -> unit
  = "soundml_amd_mfcc_bc" "soundml_amd_mfcc"

external spectral_c :
  int -> ('a, 'b) flat -> ('a, 'b) flat -> int -> int -> int -> float -> float ->
  (float, Bigarray.float64_elt) flat -> ('a, 'b) flat -> int -> unit
  = "soundml_amd_spectral_bc" "soundml_amd_spectral"

external chroma_config_c : int -> float -> float -> float -> bool -> int -> int -> chroma_handle
  = "soundml_amd_chroma_config_bc" "soundml_amd_chroma_config"

external chroma_apply_c : chroma_handle -> ('a, 'b) flat -> ('a, 'b) flat -> int -> int -> int -> int -> float -> unit
  = "soundml_amd_chroma_apply_bc" "soundml_amd_chroma_apply"

(* x, out; lead, n, power, norm kind (0 none, 1 inf, 2 p), p *)
external chroma_stft_c :
  stft_handle -> chroma_handle -> ('a, 'b) flat -> ('a, 'b) flat -> int -> int -> float -> int -> float -> unit
  = "soundml_amd_chroma_stft_bc" "soundml_amd_chroma_stft"

external to_db_c : bool -> ('a, 'b) flat -> ('a, 'b) flat -> float -> float -> float -> unit
  = "soundml_amd_to_db_bc" "soundml_amd_to_db"

(* wide (float64 chunks), channels, max_block, power (-1. = the complex face) *)
external kernel_prepare_c : stft_handle -> bool -> int -> int -> float -> kernel_handle = "soundml_amd_kernel_prepare"
external kernel_frame_bound_c : kernel_handle -> int = "soundml_amd_kernel_frame_bound"

(* chunk, out [channels; bins; capacity]; channels, m, capacity, is_flush -> frames emitted *)
external kernel_step_c : kernel_handle -> ('a, 'b) flat -> ('c, 'd) flat -> int -> int -> int -> bool -> int
  = "soundml_amd_kernel_step_bc" "soundml_amd_kernel_step"

external kernel_reset_c : kernel_handle -> unit = "soundml_amd_kernel_reset"
external stage_numbers_c : stft_handle -> int -> int * int = "soundml_amd_stage_numbers"

external fir_plan_c : (float, Bigarray.float64_elt) flat -> fir_handle = "soundml_amd_fir_plan"

external fir_apply_c :
  fir_handle -> (float, Bigarray.float32_elt) flat -> (float, Bigarray.float32_elt) flat -> int -> int -> unit
  = "soundml_amd_fir_apply"

(* the signature of the reference's [resample_shape_c] (resample.ml:1186-1196), same argument order *)
external resample_shape_c :
  (Complex.t, Bigarray.complex64_elt) flat -> (Complex.t, Bigarray.complex64_elt) flat ->
  (Complex.t, Bigarray.complex64_elt) flat -> int -> int -> int -> int -> unit
  = "soundml_amd_resample_shape_bc" "soundml_amd_resample_shape"

external resample_stage_c : (float, Bigarray.float64_elt) flat -> int -> int -> int -> stage_handle
  = "soundml_amd_resample_stage"

external resample_stage_apply_c :
  stage_handle -> (float, Bigarray.float32_elt) flat -> (float, Bigarray.float32_elt) flat -> int -> int -> unit
  = "soundml_amd_resample_stage_apply"

type rkernel_handle

external resample_kernel_prepare_c : stage_handle -> int -> int -> rkernel_handle = "soundml_amd_resample_kernel_prepare"

external resample_kernel_bounds_c : rkernel_handle -> int -> int * int = "soundml_amd_resample_kernel_bounds"

external resample_kernel_step_c :
  rkernel_handle -> (float, Bigarray.float32_elt) flat -> (float, Bigarray.float32_elt) flat -> int -> int -> int -> bool -> int
  = "soundml_amd_resample_kernel_step_bc" "soundml_amd_resample_kernel_step"

external resample_kernel_reset_c : rkernel_handle -> unit = "soundml_amd_resample_kernel_reset"


(* ---- helpers ---------------------------------------------------------------------------------------------------- *)

(* Result tensors of the offline faces.  [Nx.empty] gives fresh pageable memory: every page of a large result faults while the
   library copies into it, and the bytes cross host memory twice (C2 power_spectrum: 25-36 ms per call).  A block of the
   library's page-locked pool is written by the device directly (23.4-24.5 ms).  OPT-IN, [set_pinned_results true], because the
   block is out-of-heap memory with one owner:
     - the owner is a custom value that reports the block's bytes to the GC ([caml_alloc_custom_mem]) and returns the block to
       the pool exactly once;
     - the Bigarray over it is [CAML_BA_EXTERNAL] (no proxy, no reference count).  The owner is kept reachable by a finaliser
       closure on the one Bigarray handed to [Nx.of_buffer], i.e. for as long as the tensor is.  A Bigarray header derived from
       that one OUTSIDE Nx ([Array1.sub], [reshape], [genarray_of_array1], a view taken with [Nx.to_bigarray]) does not keep
       the owner alive: with pinned results on, copy such a view ([Nx.copy]) before the tensor goes out of scope;
     - when the pool has no page-locked memory to give, the result is an ordinary [Nx.empty].
   ([Nx_buffer.of_bigarray1] is the inverse of the [to_bigarray1] used below; [Nx.of_buffer] is the zero-copy view
   soundml-io builds its decode results with, soundml_io.ml:848-849.) *)
type host_block

external host_block_c : int -> host_block = "soundml_amd_host_block"
external host_block_array_c : host_block -> int -> int -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t = "soundml_amd_host_block_array"
external host_block_release_c : host_block -> unit = "soundml_amd_host_block_release"

let pinned_result_min_bytes = 32 * 1024 * 1024
let pinned_results = ref false
let set_pinned_results on = pinned_results := on

(* Several GPUs from this one process: [set_devices [|0; 1; 2; 3; 4; 5; 6; 7|]] makes every offline face below that takes a host
   tensor ([transform], [transform_range], [power_spectrum], [power_range], [invert], [mel_spectrogram]) split its leading axes
   into contiguous clip ranges, one range per listed device, each on a host thread of its own with that device's staging rings
   and PCIe link; every range writes its slice of the one result tensor (the per-slice law, stft_grid.ml:180-205: bit-equal to
   the single-device call).  [set_devices [||]]: the single device again.  The runtime lock is released around the call as for
   every other stub, so other domains run meanwhile. *)
external set_devices : int array -> unit = "soundml_amd_set_devices"
external device_count : unit -> int = "soundml_amd_device_count"

(* (the stub takes the Bigarray kind as its C enumerator: CAML_BA_FLOAT32 = 0, FLOAT64 = 1, COMPLEX32 = 10, COMPLEX64 = 11) *)
let kind_code_and_size : type a b. (a, b) Bigarray.kind -> int * int = function
  | Bigarray.Float32 -> (0, 4)
  | Bigarray.Float64 -> (1, 8)
  | Bigarray.Complex32 -> (10, 8)
  | Bigarray.Complex64 -> (11, 16)
  | _ -> (-1, 0)

let result_tensor dtype shape =
  let n = Array.fold_left ( * ) 1 shape in
  let kind_code, elem = kind_code_and_size (Nx_buffer.kind_of_dtype dtype) in
  if (not !pinned_results) || kind_code < 0 || n * elem < pinned_result_min_bytes then Nx.empty dtype shape
  else
    match host_block_c (n * elem) with
    | exception Failure _ -> Nx.empty dtype shape (* no page-locked memory: the staged path into an ordinary tensor *)
    | block ->
        let ba = host_block_array_c block kind_code n in
        (* the closure keeps [block] reachable while [ba] is; releasing early returns the memory at the tensor's death
           instead of at the owner's own (later) finalisation *)
        Gc.finalise (fun _ -> host_block_release_c block) ba ;
        Nx.of_buffer (Nx_buffer.of_bigarray1 ba) ~shape

let flat t = Nx_buffer.to_bigarray1 (Nx.to_buffer (Nx.contiguous t))
let flat_out t = Nx_buffer.to_bigarray1 (Nx.to_buffer t) (* freshly allocated, contiguous by construction *)
let product = Array.fold_left ( * ) 1

let split_last x =
  let shape = Nx.shape x in
  let nd = Array.length shape in
  (Array.sub shape 0 (nd - 1), shape.(nd - 1))

let split_last2 s =
  let shape = Nx.shape s in
  let nd = Array.length shape in
  (Array.sub shape 0 (nd - 2), shape.(nd - 2), shape.(nd - 1))

let alignment_code = function `Centered -> 0 | `Left -> 1 | `Right -> 2
let pad_code = function `Reflect -> (0, 0.) | `Constant v -> (1, v) | `Edge -> (2, 0.)
let scale_code = function `None -> 0 | `Magnitude -> 1 | `Psd -> 2

(* ONE handle per configuration value.  Configs are immutable and shareable (stft.mli:436-437), so the handle -- and
   with it the device tables it builds on first use (window, twiddles: ~10 uploads) -- is cached under the config's
   physical identity in an ephemeron table: it lives exactly as long as the config does and is finalised (tables freed)
   when the config is collected.  Not domain-safe, like the reference's shared lazies (resample.ml:427): a second domain
   racing on a cold entry builds a second handle, which is harmless (handles are immutable) and collected. *)
module Config_cache (K : sig type t end) = Ephemeron.K1.Make (struct
  type t = K.t
  let equal = ( == )
  let hash = Hashtbl.hash
end)

module Stft_cache = Config_cache (struct type t = Stft.Config.t end)
module Mel_cache = Config_cache (struct type t = Mel.Config.t end)
module Chroma_cache = Config_cache (struct type t = Chroma.Config.t end)

let stft_handles : stft_handle Stft_cache.t = Stft_cache.create 16
let mel_handles : mel_handle Mel_cache.t = Mel_cache.create 16
let chroma_handles : chroma_handle Chroma_cache.t = Chroma_cache.create 16

let handle_of_config (c : Stft.Config.t) =
  match Stft_cache.find_opt stft_handles c with
  | Some h -> h
  | None ->
      let pad, pad_value = pad_code (Stft.Config.pad c) in
      (* the config's own float64 window (Window.make), so all eleven window families work unchanged *)
      let window =
        flat (Window.make Nx.float64 ~periodic:true (Stft.Config.window c) (Stft.Config.win_length c))
      in
      let h =
        stft_config_c (Stft.Config.fft_size c) (Stft.Config.win_length c) (Stft.Config.hop c)
          (alignment_code (Stft.Config.alignment c)) pad pad_value (scale_code (Stft.Config.scale c)) window
      in
      Stft_cache.replace stft_handles c h ;
      h

let handle_of_mel (m : Mel.Config.t) =
  match Mel_cache.find_opt mel_handles m with
  | Some h -> h
  | None ->
      let open Mel.Config in
      let h =
        mel_config_c (n_mels m) (sample_rate m) (fft_size m) (f_min m) (f_max m)
          (match scale m with `Slaney -> 0 | `Htk -> 1)
          (match norm m with `Slaney -> 0 | `None -> 1)
      in
      Mel_cache.replace mel_handles m h ;
      h

let handle_of_chroma (c : Chroma.Config.t) =
  match Chroma_cache.find_opt chroma_handles c with
  | Some h -> h
  | None ->
      let open Chroma.Config in
      let h =
        chroma_config_c (n_chroma c) (tuning c) (ctroct c)
          (match octwidth c with Some w -> w | None -> -1.)
          (base_c c) (sample_rate c) (fft_size c)
      in
      Chroma_cache.replace chroma_handles c h ;
      h

(* ---- Stft: offline faces ------------------------------------------------------------------------------------------ *)

(* Replaces stft.ml:652-666 [transform_range] after its own range check, and through it stft.ml:632-650 [transform]:
   frames [p0, p1) of the whole signal in one device pass (borders included) instead of Kernel.step + flush + concat. *)
let transform_range cdtype (c : Stft.Config.t) ~p0 ~p1 x =
  let batch, n = split_last x in
  let lead = product batch in
  let out = result_tensor cdtype (Array.append batch [|Stft.Config.bins c; p1 - p0|]) in
  if lead > 0 && p1 > p0 then stft_range_c (handle_of_config c) (flat x) (flat_out out) lead n p0 p1 0 0. ;
  out

let transform cdtype (c : Stft.Config.t) x =
  let _, n = split_last x in
  transform_range cdtype c ~p0:0 ~p1:(Stft.frames c ~n) x

(* Replaces stft.ml:687-691 [power_spectrum]: |STFT|^power without materialising the complex spectrum. *)
let power_range ?(power = 2.) (c : Stft.Config.t) ~p0 ~p1 x =
  let batch, n = split_last x in
  let lead = product batch in
  let out = result_tensor (Nx.dtype x) (Array.append batch [|Stft.Config.bins c; p1 - p0|]) in
  if lead > 0 && p1 > p0 then stft_range_c (handle_of_config c) (flat x) (flat_out out) lead n p0 p1 1 power ;
  out

let power_spectrum ?power (c : Stft.Config.t) x =
  let _, n = split_last x in
  power_range ?power c ~p0:0 ~p1:(Stft.frames c ~n) x

(* Replaces the body of stft.ml:902-939 [synthesise] under [invert]: [check_synthesis] stays in OCaml (the C side repeats
   it with the same messages), the output is allocated here. *)
let invert dtype (c : Stft.Config.t) ?length z =
  let batch, bins, frames = split_last2 z in
  let lead = product batch in
  let out_len = match length with Some n -> n | None -> Stft.output_length c ~frames in
  let out = Nx.zeros dtype (Array.append batch [|out_len|]) in
  if lead > 0 && out_len > 0 then
    stft_invert_c (handle_of_config c) (flat z) (flat_out out) lead bins frames
      (match length with Some n -> n | None -> -1) ;
  out

(* Replaces the loop of stft.ml:961-1017 [griffin_lim] after its argument checks: 32 synthesis + analysis pairs without
   the complex spectrum ever reaching the host. *)
let griffin_lim ?(n_iter = 32) ?(momentum = 0.99) ?(init = `Zero_phase) ?length (c : Stft.Config.t) s =
  let batch, bins, frames = split_last2 s in
  let lead = product batch in
  let out_len = match length with Some n -> n | None -> Stft.output_length c ~frames in
  let out = Nx.zeros (Nx.dtype s) (Array.append batch [|out_len|]) in
  let none = Bigarray.Array1.sub (flat s) 0 0 in
  let init = match init with `Zero_phase -> none | `Phase p -> flat p in
  if lead > 0 && out_len > 0 then
    griffin_lim_c (handle_of_config c) (flat s) init (flat_out out) lead bins frames n_iter momentum
      (match length with Some n -> n | None -> -1) ;
  out

(* ---- Stft.Kernel and power_stage: the streaming face ------------------------------------------------------------- *)

(* Replaces stft.ml:597-622.  The state machine of stft.ml:366-595 (carry, prelude, tail, skip) runs in the library
   with its buffers in device memory; a step uploads the chunk and downloads the frames that completed.  Partition
   invariance (stft_law.ml:79-164) holds bit for bit: every frame is computed by the same kernel whatever the
   chunking. *)
module Kernel = struct
  type ('a, 'c) t =
    { k: kernel_handle
    ; cdtype: (Complex.t, 'c) Nx.dtype
    ; dtype: (float, 'a) Nx.dtype
    ; bins: int
    ; capacity: int
    ; mutable leading: int array }

  let prepare cdtype cfg dtype ~channels ~max_block =
    (* the reference's two checks and messages are repeated by smx_stft_kernel_prepare *)
    let wide = Nx.dtype_equal dtype Nx.float64 in
    let k = kernel_prepare_c (handle_of_config cfg) wide channels max_block (-1.) in
    {k; cdtype; dtype; bins= Stft.Config.bins cfg; capacity= kernel_frame_bound_c k; leading= [|channels|]}

  let emit t ~channels ~emitted out =
    if emitted = 0 then None
    else
      Some
        (Nx.reshape
           (Array.append t.leading [|t.bins; emitted|])
           (Nx.shrink [|(0, channels); (0, t.bins); (0, emitted)|] out) )

  let step t chunk =
    let batch, m = split_last chunk in
    let channels = product batch in
    (* the stub re-shapes a fresh kernel to the chunk's channel count and raises Invalid_argument when a stream that
       already holds samples is fed another one; [out] below is sized from the same count, which the stub checks *)
    t.leading <- batch ;
    let out = Nx.empty t.cdtype [|channels; t.bins; t.capacity|] in
    let emitted = kernel_step_c t.k (flat chunk) (flat_out out) channels m t.capacity false in
    emit t ~channels ~emitted out

  let flush t =
    let channels = product t.leading in
    let out = Nx.empty t.cdtype [|channels; t.bins; t.capacity|] in
    let none = Bigarray.Array1.sub (flat_out out) 0 0 in
    let emitted = kernel_step_c t.k none (flat_out out) channels 0 t.capacity true in
    emit t ~channels ~emitted out

  let reset t = kernel_reset_c t.k
end

(* The body of stft.ml:1364-1409 [power_stage]: the same Pipeline.kernel wiring, with [step] / [flush] emitting
   |spectrum|^power straight from the device (no complex spectrum, no [magnitude_pow] pass); latency and bound come from
   the library's restatement of stft.ml:1307-1317 and equal the reference's. *)
let power_stage_kernel ?(power = 2.) (c : Stft.Config.t) dtype ~channels ~max_block =
  let wide = Nx.dtype_equal dtype Nx.float64 in
  let k = kernel_prepare_c (handle_of_config c) wide channels max_block power in
  let capacity = kernel_frame_bound_c k and bins = Stft.Config.bins c in
  let run chunk m is_flush =
    let out = Nx.empty dtype [|channels; bins; capacity|] in
    let emitted = kernel_step_c k chunk (flat_out out) channels m capacity is_flush in
    if emitted = 0 then None else Some (Nx.shrink [|(0, channels); (0, bins); (0, emitted)|] out)
  in
  let step chunk = run (flat chunk) (snd (split_last chunk)) false in
  let flush () = run (Bigarray.Array1.create (Nx_buffer.kind_of_dtype dtype) Bigarray.c_layout 0) 0 true in
  (step, flush, fun () -> kernel_reset_c k)

(* ---- Stft.Synthesis (stft.ml:1271-1298) and synthesis_stage's body (stft.ml:1417-1442) --------------------------------
   The same signatures and messages; the kernel's state is on the device.  The library checks the prepared channel
   count against every chunk, so a stream keeps the leading shape it was prepared for (the reference fails in its
   concatenation when a later chunk disagrees). *)
type synthesis_handle

external synthesis_prepare_c : stft_handle -> bool -> int -> int -> synthesis_handle = "soundml_amd_synthesis_prepare"

external synthesis_numbers_c : stft_handle -> synthesis_handle -> int * int = "soundml_amd_synthesis_numbers"

external synthesis_step_c :
  synthesis_handle -> ('a, 'b) flat -> ('c, 'd) flat -> int -> int -> int -> int -> bool -> bool -> int
  = "soundml_amd_synthesis_step_bc" "soundml_amd_synthesis_step"

external synthesis_reset_c : synthesis_handle -> unit = "soundml_amd_synthesis_reset"

module Synthesis = struct
  type ('a, 'c) t =
    { s: synthesis_handle
    ; sdtype: (float, 'a) Nx.dtype
    ; cdtype: (Complex.t, 'c) Nx.dtype
    ; wide: bool  (* the kernel computes in float64 / complex128 (float64 audio) or float32 / complex64 *)
    ; channels: int
    ; bins: int
    ; hop: int
    ; capacity: int  (* what one step of [max_block] frames or the drain can release, plus a hop *) }

  let prepare dtype cfg (cdtype : (Complex.t, 'c) Nx.dtype) ~channels ~max_block =
    (* the reference's checks and messages (channels, max_block, check_invertible "prepare") are smx_stft_synthesis_prepare's *)
    let wide = Nx.dtype_equal dtype Nx.float64 in
    let s = synthesis_prepare_c (handle_of_config cfg) wide channels max_block in
    let _, bound = synthesis_numbers_c (handle_of_config cfg) s in
    {s; sdtype= dtype; cdtype; wide; channels; bins= Stft.Config.bins cfg; hop= Stft.Config.hop cfg; capacity= bound + Stft.Config.hop cfg}

  let emit t ~emitted out =
    if emitted = 0 then None else Some (Nx.shrink [|(0, t.channels); (0, emitted)|] out)

  (* The reference types the audio ('a) and the frames ('c) independently and converts z itself (to_complex128,
     stft.ml:1180-1200): here z is cast to the complex type the kernel was prepared for -- complex64 beside float32 audio,
     complex128 beside float64 -- so that the library never reads a buffer of another element size, and the chunk's leading
     axes must hold exactly the prepared channels (the library reads channels x bins rows of k frames). *)
  let step t z =
    let nd = Nx.ndim z in
    if nd < 2 then invalid_arg "step: cannot invert a tensor without bin and frame axes" ;
    let k = (Nx.shape z).(nd - 1) and bins = (Nx.shape z).(nd - 2) in
    let lead = Array.fold_left ( * ) 1 (Array.sub (Nx.shape z) 0 (nd - 2)) in
    if lead <> t.channels then
      invalid_arg "step: cannot feed a chunk whose leading axes disagree with the kernel's channels" ;
    let capacity = Stdlib.max t.capacity ((k * t.hop) + t.hop) in
    let out = Nx.empty t.sdtype [|t.channels; capacity|] in
    let emitted =
      if t.wide then synthesis_step_c t.s (flat (Nx.cast Nx.complex128 z)) (flat_out out) t.channels bins k capacity false true
      else synthesis_step_c t.s (flat (Nx.cast Nx.complex64 z)) (flat_out out) t.channels bins k capacity false false
    in
    emit t ~emitted out

  let flush t =
    let out = Nx.empty t.sdtype [|t.channels; t.capacity|] in
    let none = Bigarray.Array1.sub (flat_out out) 0 0 in
    let emitted = synthesis_step_c t.s none (flat_out out) t.channels t.bins 0 t.capacity true t.wide in
    emit t ~emitted out

  let reset t = synthesis_reset_c t.s
end

let stage_latency c = fst (stage_numbers_c (handle_of_config c) 1)
let frame_bound c ~max_items = snd (stage_numbers_c (handle_of_config c) max_items)

(* ---- Mel, mel_spectrogram, mfcc ------------------------------------------------------------------------------------ *)

(* Replaces the last line of mel.ml:202-231 [apply] -- [Nx.cast dtype (Nx.matmul weights (Nx.cast f64 s))] -- after its
   shape checks: the banded fp32-MFMA product on the device (float64 spectrograms: float64 dot products). *)
let mel_apply (m : Mel.Config.t) s =
  let batch, bins, frames = split_last2 s in
  let lead = product batch in
  let out = Nx.zeros (Nx.dtype s) (Array.append batch [|Mel.Config.n_mels m; frames|]) in
  if lead > 0 && frames > 0 then mel_apply_c (handle_of_mel m) (flat s) (flat_out out) lead bins frames ;
  out

(* Replaces soundml.ml:22-24 after [check_fft_sizes]: audio -> mel spectrogram in one fused launch (the power
   spectrogram stays in LDS as the MFMA's B operand; only [n_mels; frames] reaches memory). *)
let mel_spectrogram (sc : Stft.Config.t) (mc : Mel.Config.t) ?(power = 2.) x =
  let batch, n = split_last x in
  let lead = product batch in
  let frames = Stft.frames sc ~n in
  let out = Nx.zeros (Nx.dtype x) (Array.append batch [|Mel.Config.n_mels mc; frames|]) in
  if lead > 0 && frames > 0 then
    mel_spectrogram_c (handle_of_config sc) (handle_of_mel mc) (flat x) (flat_out out) lead n power ;
  out

(* Replaces soundml.ml:50-95 after its checks (fft sizes, n_mfcc range, lifter): mel spectrogram, power_to_db under the
   tensor's maximum, orthonormal DCT-II and lifter on the device. *)
let mfcc (sc : Stft.Config.t) (mc : Mel.Config.t) ?(n_mfcc = 20) ?lifter x =
  let batch, n = split_last x in
  let lead = product batch in
  let frames = Stft.frames sc ~n in
  let out = Nx.zeros (Nx.dtype x) (Array.append batch [|n_mfcc; frames|]) in
  if lead > 0 && frames > 0 then
    mfcc_c (handle_of_config sc) (handle_of_mel mc) (flat x) (flat_out out) lead n n_mfcc
      (match lifter with Some l -> l | None -> -1.) ;
  out

(* ---- Spectral.* (spectral.ml:171-255) and Chroma (chroma.ml:285-317, soundml.ml:97-107) ----------------------------- *)

let no_freqs = Bigarray.Array1.create Bigarray.float64 Bigarray.c_layout 0

(* Replaces the tail of spectral.ml:171-177 [centroid] after its own checks ([check_rank], [grid]'s shape checks): the
   reduction, the non-negativity check (same message, raised from C) and the cast.  bandwidth / rolloff / flatness
   differ only in the tag and the two scalars. *)
let spectral_feature tag ?(a = 0.) ?(b = 0.) ?freqs ?centroid ~sample_rate s =
  let batch, bins, frames = split_last2 s in
  let lead = product batch in
  let out = Nx.zeros (Nx.dtype s) (Array.append batch [|1; frames|]) in
  let fq = match freqs with Some f -> flat (Nx.cast Nx.float64 f) | None -> no_freqs in
  let cen = match centroid with Some t -> flat t | None -> Bigarray.Array1.sub (flat s) 0 0 in
  if lead > 0 && bins > 0 && frames > 0 then spectral_c tag (flat s) (flat_out out) lead bins frames a b fq cen sample_rate ;
  out

let spectral_centroid ?freqs ~sample_rate s = spectral_feature 0 ?freqs ~sample_rate s
let spectral_bandwidth ?(p = 2.) ?freqs ?centroid ~sample_rate s = spectral_feature 1 ~a:p ?freqs ?centroid ~sample_rate s
let spectral_rolloff ?(roll_percent = 0.85) ?freqs ~sample_rate s = spectral_feature 2 ~a:roll_percent ?freqs ~sample_rate s
let spectral_flatness ?(amin = 1e-10) ?(power = 2.) s = spectral_feature 3 ~a:amin ~b:power ~sample_rate:1 s

let norm_code = function `None -> (0, 0.) | `Inf -> (1, 0.) | `P p -> (2, p)

(* Replaces chroma.ml:300-317 [apply] after [check_norm]: projection, per-frame norm and the cast in one call. *)
let chroma_apply ?(norm = `Inf) (c : Chroma.Config.t) s =
  let batch, bins, frames = split_last2 s in
  let lead = product batch in
  let out = Nx.zeros (Nx.dtype s) (Array.append batch [|Chroma.Config.n_chroma c; frames|]) in
  let kind, p = norm_code norm in
  if lead > 0 && frames > 0 then chroma_apply_c (handle_of_chroma c) (flat s) (flat_out out) lead bins frames kind p ;
  out

(* Replaces soundml.ml:97-107 [chroma_stft] after its fft-size check: fused power spectrogram + projection + norm. *)
let chroma_stft (sc : Stft.Config.t) (cc : Chroma.Config.t) ?(power = 2.) ?(norm = `Inf) x =
  let batch, n = split_last x in
  let lead = product batch in
  let frames = Stft.frames sc ~n in
  let out = Nx.zeros (Nx.dtype x) (Array.append batch [|Chroma.Config.n_chroma cc; frames|]) in
  let kind, p = norm_code norm in
  if lead > 0 && frames > 0 then
    chroma_stft_c (handle_of_config sc) (handle_of_chroma cc) (flat x) (flat_out out) lead n power kind p ;
  out

(* ---- Convert.power_to_db / amplitude_to_db (convert.ml:52-62) ------------------------------------------------------- *)

(* Replaces the body of convert.ml:30-50 [to_db] after the three [check_*] calls: the floor, the logarithm, the
   reference offset and the clamp under the tensor's maximum in one call, in the tensor's own dtype. *)
let to_db ~amplitude ~reference ~amin ~top_db s =
  if Nx.numel s = 0 then Nx.copy s
  else begin
    let out = Nx.empty (Nx.dtype s) (Nx.shape s) in
    to_db_c amplitude (flat s) (flat_out out) reference amin (match top_db with Some r -> r | None -> -1.) ;
    out
  end

(* ---- FIR block convolution (BASELINE config 4; the reference lists an Effects.Filter FIR as planned) ----------------- *)

module Fir = struct
  type t = fir_handle

  let plan taps = fir_plan_c (flat (Nx.cast Nx.float64 taps))

  (* y[c][i] = sum_k h[k] x[c][i - k], zeros before the stream start; leading axes broadcast *)
  let apply (p : t) x =
    let batch, n = split_last x in
    let channels = product batch in
    let out = Nx.zeros Nx.float32 (Nx.shape x) in
    if channels > 0 && n > 0 then fir_apply_c p (flat x) (flat_out out) channels n ;
    out
end

(* ---- Resample: the two numeric pieces of the overlap-save executor --------------------------------------------------- *)

(* [resample_shape_c] above has the reference's signature: in resample.ml the only change is the external's two names
   (resample.ml:1196: "soundml_resample_shape_bc" "soundml_resample_shape" -> "soundml_amd_resample_shape_bc"
   "soundml_amd_resample_shape"); [ols_run]'s [transform_spec] (resample.ml:1500-1512) calls it unchanged.

   A whole stage on the device -- what [ols_run] + [drain] emit for an offline [apply] of a single-stage plan
   (resample.ml:1456-1599, 1745-1755): ceil (n L / M) outputs per channel, float32 interior. *)
module Resample_stage = struct
  type t = {h: stage_handle; l: int; m: int}

  let create ~proto ~l ~m ~k = {h= resample_stage_c (flat (Nx.cast Nx.float64 proto)) l m k; l; m}

  let apply (s : t) x =
    let batch, n = split_last x in
    let channels = product batch in
    let n_out = ((n * s.l) + s.m - 1) / s.m in
    let out = Nx.zeros Nx.float32 (Array.append batch [|n_out|]) in
    if channels > 0 && n_out > 0 then resample_stage_apply_c s.h (flat x) (flat_out out) channels n ;
    out
end

(* [Resample.Kernel] (resample.mli:270-319) of one pure xL or /M stage: what [ols_run] does call after call, with the block
   carry on the device.  [step] is the run of completed blocks or [None]; [flush] the virtual-silence tail or [None]; every
   partition of a signal totals [Resample_stage.apply] bit for bit. *)
module Resample_kernel = struct
  type t = {h: rkernel_handle; stage: Resample_stage.t; channels: int; max_block: int}

  let prepare (stage : Resample_stage.t) ~channels ~max_block =
    {h= resample_kernel_prepare_c stage.Resample_stage.h channels max_block; stage; channels; max_block}

  let emit (k : t) x n is_flush =
    let bound, pending = resample_kernel_bounds_c k.h n in
    let capacity = Stdlib.max 1 (if is_flush then pending else bound) in
    let out = Nx.zeros Nx.float32 [|k.channels; capacity|] in
    let got = resample_kernel_step_c k.h x (flat_out out) k.channels n capacity is_flush in
    if got = 0 then None else Some (Nx.contiguous (Nx.shrink [|(0, k.channels); (0, got)|] out))

  let step (k : t) chunk =
    let batch, n = split_last chunk in
    if product batch <> k.channels then
      invalid_arg "step: cannot feed a chunk whose leading axes disagree with the kernel's channels" ;
    emit k (flat chunk) n false

  let flush (k : t) = emit k (flat (Nx.zeros Nx.float32 [|1|])) 0 true

  let reset (k : t) = resample_kernel_reset_c k.h
end
