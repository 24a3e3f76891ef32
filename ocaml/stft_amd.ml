(* stft_amd.ml -- the OCaml side of the drop-in: the three bodies of stft.ml / mel.ml that the
   hot path funnels through, re-expressed over the C ABI.  Everything else in Soundml.Stft
   (Config, frame grid, Kernel state machine, Pipeline stages, synthesis) stays as it is.

   NOT compiled in this repository (no OCaml toolchain in the build image); see INTEGRATION.md
   for where each definition replaces the reference's. *)

type stft_handle
type mel_handle

external stft_config_c :
  int -> int -> int -> int -> int -> float -> int ->
  (float, Bigarray.float64_elt, Bigarray.c_layout) Bigarray.Array1.t -> stft_handle
  = "soundml_amd_stft_config_bc" "soundml_amd_stft_config"

(* x, out: flat Bigarray views of contiguous nx storage; lead, n, p0, p1, mode (0 complex, 1 power), power *)
external stft_range_c :
  stft_handle -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t ->
  ('c, 'd, Bigarray.c_layout) Bigarray.Array1.t -> int -> int -> int -> int -> int -> float -> unit
  = "soundml_amd_stft_range_bc" "soundml_amd_stft_range"

external mel_config_c : int -> int -> int -> float -> float -> int -> int -> mel_handle
  = "soundml_amd_mel_config_bc" "soundml_amd_mel_config"

external mel_apply_c :
  mel_handle -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t ->
  ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t -> int -> int -> int -> unit
  = "soundml_amd_mel_apply_bc" "soundml_amd_mel_apply"

external mel_spectrogram_c :
  stft_handle -> mel_handle -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t ->
  ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t -> int -> int -> float -> unit
  = "soundml_amd_mel_spectrogram_bc" "soundml_amd_mel_spectrogram"

let flat t = Nx_buffer.to_bigarray1 (Nx.to_buffer (Nx.contiguous t))

let alignment_code = function `Centered -> 0 | `Left -> 1 | `Right -> 2
let pad_code = function `Reflect -> (0, 0.) | `Constant v -> (1, v) | `Edge -> (2, 0.)
let scale_code = function `None -> 0 | `Magnitude -> 1 | `Psd -> 2

(* One handle per Stft.Config.t, built lazily from the config's own fields; the window table is
   the config's float64 window (Window.make), so all eleven window families work unchanged. *)
let handle_of_config (c : Stft.Config.t) =
  let pad, pad_value = pad_code (Stft.Config.pad c) in
  let window =
    Nx_buffer.to_bigarray1
      (Nx.to_buffer (Window.make Nx.float64 ~periodic:true (Stft.Config.window c) (Stft.Config.win_length c)))
  in
  stft_config_c (Stft.Config.fft_size c) (Stft.Config.win_length c) (Stft.Config.hop c)
    (alignment_code (Stft.Config.alignment c)) pad pad_value (scale_code (Stft.Config.scale c)) window

(* Replaces stft.ml:632-650 [transform] for the offline face: frames [0, frames) of the whole
   signal in one device pass (borders included), instead of Kernel.step + Kernel.flush + concat. *)
let transform cdtype (c : Stft.Config.t) x =
  let shape = Nx.shape x in
  let nd = Array.length shape in
  let n = shape.(nd - 1) in
  let lead = Array.fold_left ( * ) 1 (Array.sub shape 0 (nd - 1)) in
  let count = Stft.frames c ~n in
  let out = Nx.empty cdtype (Array.append (Array.sub shape 0 (nd - 1)) [|Stft.Config.bins c; count|]) in
  if lead > 0 && count > 0 then
    stft_range_c (handle_of_config c) (flat x) (Nx_buffer.to_bigarray1 (Nx.to_buffer out)) lead n 0 count 0 0. ;
  out

(* Replaces stft.ml:687-691 [power_spectrum]: |STFT|^power without materialising the complex
   spectrum (the fused kernel for fft_size 2048, the generic kernels otherwise). *)
let power_spectrum ?(power = 2.) (c : Stft.Config.t) x =
  let shape = Nx.shape x in
  let nd = Array.length shape in
  let n = shape.(nd - 1) in
  let lead = Array.fold_left ( * ) 1 (Array.sub shape 0 (nd - 1)) in
  let count = Stft.frames c ~n in
  let out =
    Nx.empty (Nx.dtype x) (Array.append (Array.sub shape 0 (nd - 1)) [|Stft.Config.bins c; count|])
  in
  if lead > 0 && count > 0 then
    stft_range_c (handle_of_config c) (flat x) (Nx_buffer.to_bigarray1 (Nx.to_buffer out)) lead n 0 count 1 power ;
  out

external stft_invert_c :
  stft_handle -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t ->
  ('c, 'd, Bigarray.c_layout) Bigarray.Array1.t -> int -> int -> int -> int -> unit
  = "soundml_amd_stft_invert_bc" "soundml_amd_stft_invert"

(* Replaces the body of stft.ml:902-939 [synthesise] under [invert]: [check_synthesis] stays in OCaml (the C
   side repeats it with the same messages), the output is allocated here. *)
let invert dtype (c : Stft.Config.t) ?length z =
  let shape = Nx.shape z in
  let nd = Array.length shape in
  let bins = shape.(nd - 2) and frames = shape.(nd - 1) in
  let batch = Array.sub shape 0 (nd - 2) in
  let lead = Array.fold_left ( * ) 1 batch in
  let out_len = match length with Some n -> n | None -> Stft.output_length c ~frames in
  let out = Nx.zeros dtype (Array.append batch [|out_len|]) in
  if lead > 0 && out_len > 0 then
    stft_invert_c (handle_of_config c) (flat z) (Nx_buffer.to_bigarray1 (Nx.to_buffer out)) lead bins frames
      (match length with Some n -> n | None -> -1) ;
  out

(* ---- Spectral.* (spectral.ml:171-255) and Chroma.apply (chroma.ml:285-317) ------------------------------ *)

external spectral_c :
  int -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t ->
  int -> int -> int -> float -> float ->
  (float, Bigarray.float64_elt, Bigarray.c_layout) Bigarray.Array1.t ->
  ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t -> int -> unit
  = "soundml_amd_spectral_bc" "soundml_amd_spectral"

let no_freqs = Bigarray.Array1.create Bigarray.float64 Bigarray.c_layout 0

(* Replaces the tail of spectral.ml:171-177 [centroid] after its own checks ([check_rank], [grid]'s shape checks):
   the reduction, the non-negativity check (same message, raised from C) and the cast.  bandwidth / rolloff /
   flatness differ only in the tag and the two scalars. *)
let spectral_feature tag ?(a = 0.) ?(b = 0.) ?freqs ~sample_rate s =
  let shape = Nx.shape s in
  let nd = Array.length shape in
  let bins = shape.(nd - 2) and frames = shape.(nd - 1) in
  let batch = Array.sub shape 0 (nd - 2) in
  let lead = Array.fold_left ( * ) 1 batch in
  let out = Nx.zeros (Nx.dtype s) (Array.append batch [|1; frames|]) in
  let fq = match freqs with Some f -> flat (Nx.cast Nx.float64 f) | None -> no_freqs in
  let none = Bigarray.Array1.sub (flat s) 0 0 in
  if lead > 0 && bins > 0 && frames > 0 then
    spectral_c tag (flat s) (Nx_buffer.to_bigarray1 (Nx.to_buffer out)) lead bins frames a b fq none sample_rate ;
  out

let spectral_centroid ?freqs ~sample_rate s = spectral_feature 0 ?freqs ~sample_rate s
let spectral_rolloff ?(roll_percent = 0.85) ?freqs ~sample_rate s = spectral_feature 2 ~a:roll_percent ?freqs ~sample_rate s
let spectral_flatness ?(amin = 1e-10) ?(power = 2.) s = spectral_feature 3 ~a:amin ~b:power ~sample_rate:1 s

type chroma_handle

external chroma_config_c : int -> float -> float -> float -> bool -> int -> int -> chroma_handle
  = "soundml_amd_chroma_config_bc" "soundml_amd_chroma_config"

external chroma_apply_c :
  chroma_handle -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t ->
  int -> int -> int -> int -> float -> unit
  = "soundml_amd_chroma_apply_bc" "soundml_amd_chroma_apply"

(* Replaces chroma.ml:300-317 [apply] after [check_norm]: projection, per-frame norm and the cast in one call. *)
let chroma_apply ?(norm = `Inf) (c : Chroma.Config.t) s =
  let open Chroma.Config in
  let h =
    chroma_config_c (n_chroma c) (tuning c) (ctroct c)
      (match octwidth c with Some w -> w | None -> -1.)
      (base_c c) (sample_rate c) (fft_size c)
  in
  let shape = Nx.shape s in
  let nd = Array.length shape in
  let bins = shape.(nd - 2) and frames = shape.(nd - 1) in
  let batch = Array.sub shape 0 (nd - 2) in
  let lead = Array.fold_left ( * ) 1 batch in
  let out = Nx.zeros (Nx.dtype s) (Array.append batch [|n_chroma c; frames|]) in
  let kind, p = match norm with `None -> (0, 0.) | `Inf -> (1, 0.) | `P p -> (2, p) in
  chroma_apply_c h (flat s) (Nx_buffer.to_bigarray1 (Nx.to_buffer out)) lead bins frames kind p ;
  out

(* ---- Convert.power_to_db / amplitude_to_db (convert.ml:52-62) ------------------------------------------- *)

external to_db_c :
  bool -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t -> ('a, 'b, Bigarray.c_layout) Bigarray.Array1.t ->
  float -> float -> float -> unit
  = "soundml_amd_to_db_bc" "soundml_amd_to_db"

(* Replaces the body of convert.ml:30-50 [to_db] after the three [check_*] calls: the floor, the logarithm, the reference
   offset and the clamp under the tensor's maximum in one call, in the tensor's own dtype. *)
let to_db ~amplitude ~reference ~amin ~top_db s =
  if Nx.numel s = 0 then Nx.copy s
  else begin
    let out = Nx.empty (Nx.dtype s) (Nx.shape s) in
    to_db_c amplitude (flat s) (Nx_buffer.to_bigarray1 (Nx.to_buffer out)) reference amin
      (match top_db with Some r -> r | None -> -1.) ;
    out
  end
