// ipc.cpp -- gdf_ipc_parser_*: the on-wire hand-off of Arrow record batches that are already in device
// memory.
//
// Reference being replaced: src/ipc.cu:77-494.  There the schema message is decoded by the Arrow C++
// library (ipc::RecordBatchStreamReader) and dumped with Arrow's JsonWriter, and the record-batch message
// header by the generated flatbuffers accessors; both libraries are fetched at configure time and are not
// part of this image.  Here the two flatbuffers tables the path needs (Message -> Schema / RecordBatch,
// format/Message.fbs and format/Schema.fbs of the Arrow specification) are read directly: a flatbuffer is
// a root uoffset, tables with a vtable of 16-bit field offsets, vectors with a 32-bit length -- ~60 lines.
//
// Contract kept (ipc.cu:124-166, 211-268, 285-405):
//   gdf_ipc_parser_open(schema, length)        `schema` is a HOST buffer holding an encapsulated Schema message;
//   gdf_ipc_parser_open_recordbatches(h, p, n) `p` is a DEVICE pointer to an encapsulated RecordBatch message
//                                              followed by its body; only the message header is copied to the host;
//   gdf_ipc_parser_get_data / _get_data_offset device pointer / offset of the body (where buffer offsets count from);
//   gdf_ipc_parser_get_layout_json             [{"name", "length", "null_count", "dtype": {"name", "bitwidth"},
//                                              "data_buffer": {"length", "offset"}, "null_buffer": {...}}, ...] with
//                                              dtype names = Arrow's Type::type enumerators, bitwidth =
//                                              (data_buffer.length / length) * 8, two buffers per node, validity first;
//   gdf_ipc_parser_get_schema_json             {"schema": {"fields": [...]}, "dictionaries": []} in the field layout of
//                                              Arrow's integration JSON (name / nullable / type / children / dictionary);
//   failures set a flag + "ParseError: ..." message instead of throwing across the C boundary.
// Both framings are accepted: the 0.x one the reference was written against (int32 size, flatbuffer) and the
// current one (0xFFFFFFFF continuation, int32 size, flatbuffer); metadata versions V4 and V5 share the 16-byte
// Buffer struct the record-batch walk relies on (V3's 24-byte struct is rejected like the reference's
// "unsupported metadata version").  Dictionary VALUES travel in DictionaryBatch messages, which a Schema
// message does not contain, so "dictionaries" lists ids only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "gdf/gdf.h"

namespace {

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- minimal flatbuffers reader (bounds-checked) ------------------------------------------------------------
struct Buf {
  const uint8_t *p;
  size_t n;
  template <class T> T at(size_t off) const {
    if (off > n || sizeof(T) > n - off) throw ParseError("flatbuffer field outside the message");   // no wrap-around
    T v;
    std::memcpy(&v, p + off, sizeof(T));
    return v;
  }
};
struct Table {
  const Buf *b = nullptr;
  size_t pos = 0;
  bool ok() const { return b != nullptr; }
  // byte position of field `id`, or 0 when the field is absent (default value applies)
  size_t field(int id) const {
    const int32_t so = b->at<int32_t>(pos);
    const int64_t vt64 = (int64_t)pos - so;
    if (vt64 < 0 || (uint64_t)vt64 >= b->n) throw ParseError("vtable outside the message");   // a crafted soffset must not wrap
    const size_t vt = (size_t)vt64;
    const uint16_t vsize = b->at<uint16_t>(vt);
    const size_t slot = 4 + 2 * (size_t)id;
    if (slot + 2 > vsize) return 0;
    const uint16_t off = b->at<uint16_t>(vt + slot);
    return off ? pos + off : 0;
  }
  template <class T> T scalar(int id, T dflt) const { const size_t f = field(id); return f ? b->at<T>(f) : dflt; }
  size_t indirect(int id) const { const size_t f = field(id); return f ? f + b->at<uint32_t>(f) : 0; }
  Table table(int id) const { const size_t t = indirect(id); return t ? Table{b, t} : Table{}; }
  std::string str(int id) const {
    const size_t s = indirect(id);
    if (!s) return std::string();
    const uint32_t len = b->at<uint32_t>(s);
    if (s > b->n || b->n - s < 4 || len > b->n - s - 4) throw ParseError("string outside the message");
    return std::string((const char *)b->p + s + 4, len);
  }
  // vector field: position of element 0 and the element count
  size_t vec(int id, uint32_t *count) const {
    const size_t v = indirect(id);
    *count = v ? b->at<uint32_t>(v) : 0;
    return v ? v + 4 : 0;
  }
};
Table root_table(const Buf &b) { return Table{&b, (size_t)b.at<uint32_t>(0)}; }

// ---- Arrow metadata (format/Schema.fbs, format/Message.fbs) ---------------------------------------------------
enum { HDR_SCHEMA = 1, HDR_DICTIONARY = 2, HDR_RECORDBATCH = 3 };
enum { T_NULL = 1, T_INT = 2, T_FLOAT = 3, T_BINARY = 4, T_UTF8 = 5, T_BOOL = 6, T_DECIMAL = 7, T_DATE = 8, T_TIME = 9,
       T_TIMESTAMP = 10, T_INTERVAL = 11, T_LIST = 12, T_STRUCT = 13, T_UNION = 14, T_FIXEDBINARY = 15, T_FIXEDLIST = 16,
       T_MAP = 17, T_DURATION = 18, T_LARGEBINARY = 19, T_LARGEUTF8 = 20, T_LARGELIST = 21 };

struct FieldDesc {
  std::string name;
  std::string type_name;   // Arrow Type::type enumerator, as GetTypeName() of ipc.cu:40-75
  std::string type_json;   // integration-JSON "type" object
  bool nullable = true;
  bool has_dict = false;
  int64_t dict_id = 0;
  std::string dict_index_json;
  bool dict_ordered = false;
  std::vector<FieldDesc> children;
};

std::string json_escape(const std::string &s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += c; }
    else if ((unsigned char)c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
    else o += c;
  }
  return o;
}

std::string int_json(int bits, bool is_signed) {
  std::ostringstream o;
  o << "{\"name\": \"int\", \"bitWidth\": " << bits << ", \"isSigned\": " << (is_signed ? "true" : "false") << "}";
  return o.str();
}

void describe_type(int type_type, const Table &t, FieldDesc *f) {
  static const char *units[] = {"SECOND", "MILLISECOND", "MICROSECOND", "NANOSECOND"};
  std::ostringstream js;
  switch (type_type) {
    case T_NULL: f->type_name = "NA"; js << "{\"name\": \"null\"}"; break;
    case T_INT: {
      const int bits = t.ok() ? t.scalar<int32_t>(0, 0) : 0;
      const bool sg = t.ok() ? t.scalar<uint8_t>(1, 0) != 0 : false;
      f->type_name = std::string(sg ? "INT" : "UINT") + std::to_string(bits);
      js << int_json(bits, sg);
      break;
    }
    case T_FLOAT: {
      const int prec = t.ok() ? t.scalar<int16_t>(0, 0) : 0;
      f->type_name = prec == 0 ? "HALF_FLOAT" : (prec == 1 ? "FLOAT" : "DOUBLE");
      js << "{\"name\": \"floatingpoint\", \"precision\": \"" << (prec == 0 ? "HALF" : (prec == 1 ? "SINGLE" : "DOUBLE")) << "\"}";
      break;
    }
    case T_BINARY: f->type_name = "BINARY"; js << "{\"name\": \"binary\"}"; break;
    case T_UTF8: f->type_name = "STRING"; js << "{\"name\": \"utf8\"}"; break;
    case T_BOOL: f->type_name = "BOOL"; js << "{\"name\": \"bool\"}"; break;
    case T_DECIMAL:
      f->type_name = "DECIMAL";
      js << "{\"name\": \"decimal\", \"precision\": " << (t.ok() ? t.scalar<int32_t>(0, 0) : 0) << ", \"scale\": "
         << (t.ok() ? t.scalar<int32_t>(1, 0) : 0) << "}";
      break;
    case T_DATE: {
      const int unit = t.ok() ? t.scalar<int16_t>(0, 1) : 1;      // default MILLISECOND
      f->type_name = unit == 0 ? "DATE32" : "DATE64";
      js << "{\"name\": \"date\", \"unit\": \"" << (unit == 0 ? "DAY" : "MILLISECOND") << "\"}";
      break;
    }
    case T_TIME: {
      const int unit = t.ok() ? t.scalar<int16_t>(0, 1) : 1;
      const int bits = t.ok() ? t.scalar<int32_t>(1, 32) : 32;
      f->type_name = bits == 32 ? "TIME32" : "TIME64";
      js << "{\"name\": \"time\", \"unit\": \"" << units[unit & 3] << "\", \"bitWidth\": " << bits << "}";
      break;
    }
    case T_TIMESTAMP: {
      const int unit = t.ok() ? t.scalar<int16_t>(0, 0) : 0;
      f->type_name = "TIMESTAMP";
      js << "{\"name\": \"timestamp\", \"unit\": \"" << units[unit & 3] << "\"";
      const std::string tz = t.ok() ? t.str(1) : std::string();
      if (!tz.empty()) js << ", \"timezone\": \"" << json_escape(tz) << "\"";
      js << "}";
      break;
    }
    case T_INTERVAL: f->type_name = "INTERVAL"; js << "{\"name\": \"interval\"}"; break;
    case T_LIST: case T_LARGELIST: case T_FIXEDLIST: f->type_name = "LIST"; js << "{\"name\": \"list\"}"; break;
    case T_STRUCT: f->type_name = "STRUCT"; js << "{\"name\": \"struct\"}"; break;
    case T_UNION: f->type_name = "UNION"; js << "{\"name\": \"union\"}"; break;
    case T_FIXEDBINARY:
      f->type_name = "FIXED_SIZE_BINARY";
      js << "{\"name\": \"fixedsizebinary\", \"byteWidth\": " << (t.ok() ? t.scalar<int32_t>(0, 0) : 0) << "}";
      break;
    case T_MAP: f->type_name = "MAP"; js << "{\"name\": \"map\"}"; break;
    case T_LARGEBINARY: f->type_name = "BINARY"; js << "{\"name\": \"largebinary\"}"; break;
    case T_LARGEUTF8: f->type_name = "STRING"; js << "{\"name\": \"largeutf8\"}"; break;
    default: f->type_name = "UNKNOWN"; js << "{\"name\": \"unknown\"}"; break;
  }
  f->type_json = js.str();
}

FieldDesc parse_field(const Table &t, int depth) {
  if (depth > 32) throw ParseError("schema nesting too deep");
  FieldDesc f;
  f.name = t.str(0);
  f.nullable = t.scalar<uint8_t>(1, 0) != 0;
  describe_type(t.scalar<uint8_t>(2, 0), t.table(3), &f);
  const Table d = t.table(4);
  if (d.ok()) {
    f.has_dict = true;
    f.dict_id = d.scalar<int64_t>(0, 0);
    const Table it = d.table(1);
    f.dict_index_json = int_json(it.ok() ? it.scalar<int32_t>(0, 32) : 32, it.ok() ? it.scalar<uint8_t>(1, 1) != 0 : true);
    f.dict_ordered = d.scalar<uint8_t>(2, 0) != 0;
    f.type_name = "DICTIONARY";          // Type::DICTIONARY: what the data buffer holds is the index array
  }
  uint32_t nchild = 0;
  const size_t c0 = t.vec(5, &nchild);
  for (uint32_t i = 0; i < nchild; ++i) {
    const size_t slot = c0 + 4 * (size_t)i;
    f.children.push_back(parse_field(Table{t.b, slot + t.b->at<uint32_t>(slot)}, depth + 1));
  }
  return f;
}

void field_json(std::ostream &os, const FieldDesc &f) {
  os << "{\"name\": \"" << json_escape(f.name) << "\", \"nullable\": " << (f.nullable ? "true" : "false") << ", \"type\": " << f.type_json
     << ", \"children\": [";
  for (size_t i = 0; i < f.children.size(); ++i) {
    if (i) os << ", ";
    field_json(os, f.children[i]);
  }
  os << "]";
  if (f.has_dict)
    os << ", \"dictionary\": {\"id\": " << f.dict_id << ", \"indexType\": " << f.dict_index_json << ", \"isOrdered\": "
       << (f.dict_ordered ? "true" : "false") << "}";
  os << "}";
}

// encapsulated message: [0xFFFFFFFF] int32 metadata_size, flatbuffer.  Returns the offset of the flatbuffer.
size_t message_prefix(const uint8_t *p, size_t n, int32_t *meta_size) {
  if (n < 4) throw ParseError("message shorter than its length prefix");
  size_t off = 0;
  int32_t v;
  std::memcpy(&v, p, 4);
  if ((uint32_t)v == 0xFFFFFFFFu) {
    if (n < 8) throw ParseError("message shorter than its length prefix");
    std::memcpy(&v, p + 4, 4);
    off = 4;
  }
  if (v <= 0) throw ParseError("non-positive message size");
  *meta_size = v;
  return off + 4;
}

class IpcParser {
 public:
  void open(const uint8_t *schema, size_t length) {
    guard([&] {
      if (!fields_.empty() || !nodes_.empty()) throw ParseError("cannot open more than once");
      if (!schema) throw ParseError("null schema buffer");
      int32_t msize = 0;
      const size_t fb = message_prefix(schema, length, &msize);
      if (fb + (size_t)msize > length) throw ParseError("schema message is truncated");
      const Buf b{schema + fb, (size_t)msize};
      const Table msg = root_table(b);
      if (msg.scalar<uint8_t>(1, 0) != HDR_SCHEMA) throw ParseError("expecting schema type");
      const Table sch = msg.table(2);
      if (!sch.ok()) throw ParseError("failed to parse schema");
      uint32_t nf = 0;
      const size_t f0 = sch.vec(1, &nf);
      for (uint32_t i = 0; i < nf; ++i) {
        const size_t slot = f0 + 4 * (size_t)i;
        fields_.push_back(parse_field(Table{&b, slot + b.at<uint32_t>(slot)}, 0));
      }
      std::ostringstream os;
      os << "{\"schema\": {\"fields\": [";
      for (size_t i = 0; i < fields_.size(); ++i) {
        if (i) os << ", ";
        field_json(os, fields_[i]);
      }
      os << "]}, \"dictionaries\": [";
      bool first = true;
      for (const FieldDesc &f : fields_)
        if (f.has_dict) { os << (first ? "" : ", ") << "{\"id\": " << f.dict_id << "}"; first = false; }
      os << "]}";
      schema_json_ = os.str();
    });
  }

  void open_recordbatches(const uint8_t *d_buf, size_t length) {
    guard([&] {
      if (!d_buf) throw ParseError("null record batch buffer");
      d_buffer_ = d_buf;
      uint8_t prefix[8] = {0};
      fetch(prefix, d_buf, length < 8 ? length : 8);
      int32_t msize = 0;
      const size_t fb = message_prefix(prefix, length < 8 ? length : 8, &msize);
      if (fb + (size_t)msize > length) throw ParseError("record batch message is truncated");
      std::vector<uint8_t> meta((size_t)msize);
      fetch(meta.data(), d_buf + fb, (size_t)msize);
      const Buf b{meta.data(), meta.size()};
      const Table msg = root_table(b);
      const int version = msg.scalar<int16_t>(0, 0);          // MetadataVersion: V1=0 ... V4=3, V5=4
      if (version < 3) throw ParseError("unsupported metadata version, expected V4 or V5 got V" + std::to_string(version + 1));
      if (msg.scalar<int64_t>(3, 0) <= 0) throw ParseError("recordbatch should have a body");
      if (msg.scalar<uint8_t>(1, 0) != HDR_RECORDBATCH) throw ParseError("expecting recordbatch type");
      d_body_ = d_buf + fb + (size_t)msize;
      const Table rb = msg.table(2);
      if (!rb.ok()) throw ParseError("expecting recordbatch type");
      uint32_t nnodes = 0, nbufs = 0;
      const size_t n0 = rb.vec(1, &nnodes), b0 = rb.vec(2, &nbufs);
      if ((uint64_t)nnodes * 2 != nbufs) throw ParseError("unexpected: more than 2 buffers per node!?");
      if (nnodes > fields_.size()) throw ParseError("record batch has more nodes than the schema has fields");
      std::ostringstream os;
      os << "[";
      for (uint32_t i = 0; i < nnodes; ++i) {
        Node nd;
        nd.name = fields_[i].name;
        nd.dtype = fields_[i].type_name;
        nd.length = b.at<int64_t>(n0 + 16 * (size_t)i);
        nd.null_count = b.at<int64_t>(n0 + 16 * (size_t)i + 8);
        nd.null_off = b.at<int64_t>(b0 + 16 * (size_t)(2 * i));            // first buffer: validity bitmap
        nd.null_len = b.at<int64_t>(b0 + 16 * (size_t)(2 * i) + 8);
        nd.data_off = b.at<int64_t>(b0 + 16 * (size_t)(2 * i + 1));
        nd.data_len = b.at<int64_t>(b0 + 16 * (size_t)(2 * i + 1) + 8);
        nd.bitwidth = nd.length > 0 ? (int)((nd.data_len / nd.length) * 8) : 0;   // ipc.cu:383
        nodes_.push_back(nd);
        if (i) os << ", ";
        os << "{\"name\": \"" << json_escape(nd.name) << "\", \"length\": " << nd.length << ", \"null_count\": " << nd.null_count
           << ", \"dtype\": {\"name\": \"" << nd.dtype << "\", \"bitwidth\": " << nd.bitwidth << "}, \"data_buffer\": {\"length\": "
           << nd.data_len << ", \"offset\": " << nd.data_off << "}, \"null_buffer\": {\"length\": " << nd.null_len
           << ", \"offset\": " << nd.null_off << "}}";
      }
      os << "]";
      layout_json_ = os.str();
    });
  }

  bool failed() const { return failed_; }
  const std::string &error() const { return error_; }
  const std::string &schema_json() const { return schema_json_; }
  const std::string &layout_json() { if (layout_json_.empty()) layout_json_ = "[]"; return layout_json_; }
  const void *data() const { return d_body_; }
  int64_t data_offset() const { return d_body_ && d_buffer_ ? (int64_t)(d_body_ - d_buffer_) : 0; }

 private:
  struct Node { std::string name, dtype; int64_t length, null_count, null_off, null_len, data_off, data_len; int bitwidth; };

  template <class F> void guard(F &&f) {
    try { f(); }
    catch (const std::exception &e) { error_ = std::string("ParseError: ") + e.what(); failed_ = true; }
  }
  static void fetch(void *dst, const uint8_t *d_src, size_t n) {
    if (n == 0) return;
    if (hipMemcpy(dst, d_src, n, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); throw ParseError("cannot read value"); }
  }

  std::vector<FieldDesc> fields_;
  std::vector<Node> nodes_;
  const uint8_t *d_buffer_ = nullptr, *d_body_ = nullptr;
  bool failed_ = false;
  std::string error_, schema_json_ = "{}", layout_json_;
};

IpcParser *unwrap(gdf_ipc_parser_type *h) { return reinterpret_cast<IpcParser *>(h); }

}  // namespace

extern "C" {

gdf_ipc_parser_type *gdf_ipc_parser_open(const uint8_t *schema, size_t length) {
  IpcParser *p = new (std::nothrow) IpcParser;
  if (p) p->open(schema, length);
  return reinterpret_cast<gdf_ipc_parser_type *>(p);
}
void gdf_ipc_parser_open_recordbatches(gdf_ipc_parser_type *handle, const uint8_t *recordbatches, size_t length) {
  if (handle) unwrap(handle)->open_recordbatches(recordbatches, length);
}
void gdf_ipc_parser_close(gdf_ipc_parser_type *handle) { delete unwrap(handle); }
int gdf_ipc_parser_failed(gdf_ipc_parser_type *handle) { return handle ? unwrap(handle)->failed() : 1; }
const char *gdf_ipc_parser_to_json(gdf_ipc_parser_type *handle) { return handle ? unwrap(handle)->layout_json().c_str() : "[]"; }
const char *gdf_ipc_parser_get_error(gdf_ipc_parser_type *handle) { return handle ? unwrap(handle)->error().c_str() : "null parser handle"; }
const void *gdf_ipc_parser_get_data(gdf_ipc_parser_type *handle) { return handle ? unwrap(handle)->data() : nullptr; }
int64_t gdf_ipc_parser_get_data_offset(gdf_ipc_parser_type *handle) { return handle ? unwrap(handle)->data_offset() : 0; }
const char *gdf_ipc_parser_get_schema_json(gdf_ipc_parser_type *handle) { return handle ? unwrap(handle)->schema_json().c_str() : "{}"; }
const char *gdf_ipc_parser_get_layout_json(gdf_ipc_parser_type *handle) { return handle ? unwrap(handle)->layout_json().c_str() : "[]"; }

}  // extern "C"
