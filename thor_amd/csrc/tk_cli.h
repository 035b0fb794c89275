// tk_cli.h - minimal Thorenc-compatible option/config-file reader for the test drivers.
// Same surface as enc/strings.c:287-356 for the options this path honours: "-name value" tokens,
// ';' starts a comment, "-cf file" includes a config file (command line wins over later files,
// like the reference, because explicit arguments are applied after the config contents).
#pragma once
#include <map>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>
#include <cstdlib>
#include "tk_encoder.h"

namespace tk {

struct CliArgs {
  SeqParams sp;
  std::string infile, outfile, recfile;
  int num_frames = 600, skip = 0, streams = 1;
  // Options of the reference's table (enc/strings.c:287-356) this path does not implement.  A value other than
  // the reference's default would change the bitstream, so it is recorded here and rejected by the caller
  // (thor_hip_params_set / thor_hip_params_from_config return non-zero, the CLI tools exit) - never ignored.
  std::string unsupported;   // first "-name value" that cannot be honoured
  std::string unknown;       // first option that is not in the reference's table at all
};

static inline void cli_tokens_from_file(const std::string& path, std::vector<std::string>& out) {
  std::ifstream f(path);
  if (!f) { fprintf(stderr, "cannot open config %s\n", path.c_str()); exit(2); }
  std::string line;
  while (std::getline(f, line)) {
    size_t sc = line.find(';');
    if (sc != std::string::npos) line.resize(sc);
    std::istringstream is(line);
    std::string tok;
    while (is >> tok) out.push_back(tok);
  }
}

static inline void cli_apply(CliArgs& a, const std::vector<std::string>& t) {
  for (size_t i = 0; i + 1 < t.size(); i += 2) {
    const std::string& k = t[i];
    const std::string& v = t[i + 1];
    SeqParams& p = a.sp;
    auto I = [&]() { return atoi(v.c_str()); };
    auto F = [&]() { return (float)atof(v.c_str()); };
    if (k == "-cf") { std::vector<std::string> sub; cli_tokens_from_file(v, sub); cli_apply(a, sub); }
    else if (k == "-if") a.infile = v;
    else if (k == "-of") a.outfile = v;
    else if (k == "-rf") a.recfile = v;
    else if (k == "-n") a.num_frames = I();
    else if (k == "-skip") a.skip = I();
    else if (k == "-streams") a.streams = I();
    else if (k == "-width") p.width = I();
    else if (k == "-height") p.height = I();
    else if (k == "-qp") p.qp = I();
    else if (k == "-f") p.frame_rate = F();
    else if (k == "-lambda_coeffI") p.lambda_coeffI = F();
    else if (k == "-lambda_coeffP") p.lambda_coeffP = F();
    else if (k == "-early_skip_thr") p.early_skip_thr = F();
    else if (k == "-enable_tb_split") p.enable_tb_split = I();
    else if (k == "-enable_pb_split") p.enable_pb_split = I();
    else if (k == "-max_num_ref") p.max_num_ref = I();
    else if (k == "-HQperiod") p.HQperiod = I();
    else if (k == "-num_reorder_pics") p.num_reorder_pics = I();
    else if (k == "-interp_ref") p.interp_ref = I();
    else if (k == "-max_clpf_strength") p.max_clpf_strength = I();
    else if (k == "-dqpP") p.dqpP = I();
    else if (k == "-dqpB") p.dqpB = I();
    else if (k == "-dqpB0") p.dqpB0 = I();
    else if (k == "-dqpB1") p.dqpB1 = I();
    else if (k == "-dqpB2") p.dqpB2 = I();
    else if (k == "-dqpB3") p.dqpB3 = I();
    else if (k == "-mqpB") p.mqpB = F();
    else if (k == "-mqpB0") p.mqpB0 = F();
    else if (k == "-mqpB1") p.mqpB1 = F();
    else if (k == "-mqpB2") p.mqpB2 = F();
    else if (k == "-mqpB3") p.mqpB3 = F();
    else if (k == "-lambda_coeffB") p.lambda_coeffB = F();
    else if (k == "-lambda_coeffB0") p.lambda_coeffB0 = F();
    else if (k == "-lambda_coeffB1") p.lambda_coeffB1 = F();
    else if (k == "-lambda_coeffB2") p.lambda_coeffB2 = F();
    else if (k == "-lambda_coeffB3") p.lambda_coeffB3 = F();
    else if (k == "-dyadic_coding") p.dyadic_coding = I();
    else if (k == "-dqpI") p.dqpI = I();
    else if (k == "-mqpP") p.mqpP = F();
    else if (k == "-intra_period") p.intra_period = I();
    else if (k == "-intra_rdo") p.intra_rdo = I();
    else if (k == "-encoder_speed") p.encoder_speed = I();
    else if (k == "-deblocking") p.deblocking = I();
    else if (k == "-cdef") p.cdef = I();
    else if (k == "-clpf") p.clpf = I();
    else if (k == "-use_block_contexts") p.use_block_contexts = I();
    else if (k == "-enable_bipred") p.enable_bipred = I();
    else if (k == "-enable_cfl_intra") p.cfl_intra = I();
    else if (k == "-enable_cfl_inter") p.cfl_inter = I();
    else if (k == "-bitdepth") p.bitdepth = I();
    else if (k == "-input_bitdepth") p.input_bitdepth = I();
    else {
      // the rest of the reference's table: harmless at their defaults (or never read by the block path), fatal otherwise
      struct Opt { const char* name; const char* dflt; };  // dflt == nullptr: any value is fine (I/O and reporting options)
      static const Opt rest[] = {{"-ph", "0"}, {"-fh", "0"}, {"-stat", nullptr}, {"-snrcalc", nullptr}, {"-log2_sb_size", "7"},
                                 {"-max_delta_qp", "0"}, {"-delta_qp_step", nullptr}, {"-sync", "0"}, {"-bitrate", "0"},
                                 {"-max_qp", nullptr}, {"-min_qp", nullptr}, {"-max_qpI", nullptr}, {"-min_qpI", nullptr},
                                 {"-qmtx", "0"}, {"-qmtx_offset", nullptr}, {"-subsample", "420"}, {"-frame_bitdepth", nullptr}};
      bool found = false;
      for (const Opt& o : rest)
        if (k == o.name) {
          found = true;
          if (o.dflt && atoi(v.c_str()) != atoi(o.dflt) && a.unsupported.empty()) a.unsupported = k + " " + v;
        }
      if (!found && a.unknown.empty()) a.unknown = k;
    }
  }
  if (t.size() & 1) { if (a.unknown.empty()) a.unknown = t.back() + " (no value)"; }
}

static inline CliArgs cli_parse(int argc, char** argv) {
  CliArgs a;
  std::vector<std::string> t;
  for (int i = 1; i < argc; i++) t.push_back(argv[i]);
  // config files first, explicit arguments afterwards (enc/strings.c:196-265 applies argv last)
  std::vector<std::string> files, rest;
  for (size_t i = 0; i + 1 < t.size(); i += 2) {
    if (t[i] == "-cf") { files.push_back(t[i]); files.push_back(t[i + 1]); }
    else { rest.push_back(t[i]); rest.push_back(t[i + 1]); }
  }
  cli_apply(a, files);
  cli_apply(a, rest);
  if (!a.unknown.empty()) { fprintf(stderr, "Run-time error...\nunknown option %s\n...now exiting to system...\n", a.unknown.c_str()); exit(2); }
  if (!a.unsupported.empty()) { fprintf(stderr, "Run-time error...\noption %s is not implemented by this path (it changes the bitstream)\n...now exiting to system...\n", a.unsupported.c_str()); exit(2); }
  return a;
}

// Encode `num_frames` frames of a raw 4:2:0 file with S identical-geometry streams: stream s
// codes frames [skip + s*num_frames, skip + (s+1)*num_frames) as its own closed stream (what the
// reference produces with -skip/-n).  Outputs "<of>" for S==1, "<of>.<s>" otherwise.
template <typename PIX> int cli_run(const CliArgs& a) {
  const SeqParams& p = a.sp;
  FILE* fi = fopen(a.infile.c_str(), "rb");
  if (!fi) { fprintf(stderr, "cannot open %s\n", a.infile.c_str()); return 2; }
  const size_t fsz = (size_t)p.width * p.height * 3 / 2;
  Engine<PIX> eng;
  eng.open(p, a.streams);
  std::vector<PIX> frame(fsz), rec(fsz);
  std::vector<FILE*> fr(a.streams, nullptr);
  auto name = [&](const std::string& base, int s) { return a.streams == 1 ? base : base + "." + std::to_string(s); };
  for (int s = 0; s < a.streams; s++)
    if (!a.recfile.empty()) fr[s] = fopen(name(a.recfile, s).c_str(), "wb");
  // all streams run in lock step through their (identical) coding-order schedules
  fseek(fi, 0, SEEK_END);
  const int file_frames = (int)(ftell(fi) / (long)(fsz * sizeof(PIX)));
  for (int s = 0; s < a.streams; s++) eng.begin_sequence(s, a.skip + s * a.num_frames, a.num_frames, file_frames);
  std::vector<std::vector<PIX>> recs((size_t)a.streams * a.num_frames);  // recon in display order
  // THOR_STAGGER=1 (tests): the streams in two groups half a frame apart (Engine::encode_run) instead of lock step
  if (getenv("THOR_STAGGER") && atoi(getenv("THOR_STAGGER")) && a.streams > 1) {
    int rc = 0;
    eng.encode_run(a.num_frames,
        [&](int s) -> bool {
          if (!eng.schedule(s)) return false;
          const size_t idx = (size_t)eng.st[s].cur_abs;
          if (fseek(fi, (long)(idx * fsz * sizeof(PIX)), SEEK_SET) || fread(frame.data(), sizeof(PIX), fsz, fi) != fsz) { fprintf(stderr, "short read at frame %zu\n", idx); rc = 3; return false; }
          eng.upload_orig(s, frame.data());
          return true;
        },
        [&](int first, int count) {
          for (int s = first; s < first + count; s++)
            if (fr[s]) { eng.download_rec(s, rec.data()); recs[(size_t)s * a.num_frames + eng.st[s].cur.frame_num] = rec; }
        });
    if (rc) return rc;
  } else
  for (;;) {
    std::vector<FrameParams> fp(a.streams);
    int active = 0;
    for (int s = 0; s < a.streams; s++) {
      if (!eng.schedule(s)) continue;
      active++;
      size_t idx = (size_t)eng.st[s].cur_abs;
      if (fseek(fi, (long)(idx * fsz * sizeof(PIX)), SEEK_SET) || fread(frame.data(), sizeof(PIX), fsz, fi) != fsz) {
        fprintf(stderr, "short read at frame %zu\n", idx);
        return 3;
      }
      eng.upload_orig(s, frame.data());
      fp[s] = eng.st[s].cur;
    }
    if (!active) break;
    if (active != a.streams) { fprintf(stderr, "streams out of step\n"); return 4; }
    eng.encode_frames(fp);
    if (const char* ddp = getenv("THOR_DD_DUMP")) {   // tests: stream 0's block data in the record format of oracle/dd_shim.c
      std::vector<DbCell> cells(eng.num_cells());
      eng.download_cells(0, cells.data());
      FILE* fd = fopen(ddp, "ab");
      if (!fd) { fprintf(stderr, "cannot open %s\n", ddp); return 5; }
      const int32_t hdr[3] = {0x44444444, fp[0].frame_num, (int32_t)cells.size()};
      fwrite(hdr, sizeof hdr, 1, fd);
      for (const DbCell& c : cells) {
        const DdFields d = dd_fields(c);
        const int32_t r[14] = {d.mode, d.cbp_y, d.cbp_u, d.cbp_v, d.size, d.tb_split, d.pb_part, d.mv0x, d.mv0y, d.mv1x, d.mv1y, d.ref_idx0, d.ref_idx1, d.bipred_flag};
        fwrite(r, sizeof r, 1, fd);
      }
      fclose(fd);
    }
    for (int s = 0; s < a.streams; s++)
      if (fr[s]) {
        eng.download_rec(s, rec.data());
        recs[(size_t)s * a.num_frames + fp[s].frame_num] = rec;
      }
  }
  for (int s = 0; s < a.streams; s++)
    if (fr[s])
      for (int n = 0; n < a.num_frames; n++) {
        const std::vector<PIX>& r = recs[(size_t)s * a.num_frames + n];
        if (!r.empty()) fwrite(r.data(), sizeof(PIX), fsz, fr[s]);
      }
  for (int s = 0; s < a.streams; s++) {
    if (fr[s]) fclose(fr[s]);
    if (!a.outfile.empty()) {
      FILE* fo = fopen(name(a.outfile, s).c_str(), "wb");
      fwrite(eng.st[s].out.data(), 1, eng.st[s].out.size(), fo);
      fclose(fo);
    }
  }
  eng.close();
  fclose(fi);
  return 0;
}

}  // namespace tk
