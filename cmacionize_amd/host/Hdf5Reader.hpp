/*
 * Hdf5Reader.hpp - a minimal, dependency-free reader of HDF5 files: what the
 * reference reads from a snapshot through src/HDF5Tools.hpp (open_group,
 * group_exists, get_attribute_names, read_attribute, read_dataset) - enough
 * for the files Hdf5Writer.hpp writes and for the files libhdf5 writes by
 * default for the reference's GadgetDensityGridWriter
 * (src/GadgetDensityGridWriter.cpp:107-358):
 *
 *   superblock version 0 / 1, version-1 object headers (with continuation
 *   blocks), groups as symbol tables (version-1 B-trees of any depth, local
 *   heaps), attribute messages version 1 - 3, data layout version 3: compact,
 *   contiguous, or chunked (version-1 chunk B-tree) with the deflate and
 *   shuffle filters (the reference chunks every dataset and compresses on
 *   request, src/HDF5Tools.hpp:1363-1379); little-endian integers and IEEE
 *   floats of 4 / 8 bytes, fixed-length strings.
 *
 * Not read (a clear error instead): version-2 object headers / new-style
 * groups (only written with libver=latest), variable-length data, other
 * filters. The image has no HDF5 library; the layout follows the HDF5 File
 * Format Specification version 1.1 / 2.0. zlib inflates deflated chunks.
 */
#ifndef CMI_HDF5READER_HPP
#define CMI_HDF5READER_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include <zlib.h>

namespace cmi {

class Hdf5Reader {
public:
  struct Type {
    int cls = -1; /* 0 fixed point, 1 floating point, 3 string, 6 compound */
    uint32_t size = 0;
    bool is_signed = false;
    /* a compound type's members (of the three simple classes) */
    struct Member {
      std::string name;
      uint32_t offset = 0;
      int cls = -1;
      uint32_t size = 0;
      bool is_signed = false;
    };
    std::vector<Member> members;
    size_t message_bytes = 0; /* length of the datatype message */
  };
  struct Attribute {
    Type type;
    std::vector<uint64_t> dims;
    std::vector<uint8_t> data;
  };
  struct Object {
    std::map<std::string, uint64_t> members; /* name -> header address */
    std::map<std::string, Attribute> attributes;
    bool is_group = false;
    /* a dataset */
    Type type;
    std::vector<uint64_t> dims;
    int layout = -1; /* 0 compact, 1 contiguous, 2 chunked */
    uint64_t data_address = ~(uint64_t)0, data_size = 0;
    std::vector<uint8_t> compact;
    std::vector<uint64_t> chunk_dims; /* + the element size last */
    std::vector<uint16_t> filters;
  };

private:
  static constexpr uint64_t UNDEF = ~(uint64_t)0;
  std::string _name;
  mutable std::ifstream _file;
  uint64_t _size = 0, _base = 0, _root = 0;

  [[noreturn]] void error(const std::string &what) const {
    throw std::runtime_error("HDF5 file \"" + _name + "\": " + what);
  }

  std::vector<uint8_t> bytes(uint64_t address, uint64_t n) const {
    if (address == UNDEF || address + _base + n > _size)
      error("a block of " + std::to_string(n) + " bytes at " +
            std::to_string(address) + " lies outside the file");
    std::vector<uint8_t> b(n);
    _file.seekg((std::streamoff)(address + _base));
    _file.read(reinterpret_cast<char *>(b.data()), (std::streamsize)n);
    if (!_file)
      error("short read");
    return b;
  }
  template <typename T>
  static T get(const std::vector<uint8_t> &b, size_t at) {
    T v;
    if (at + sizeof(T) > b.size())
      throw std::runtime_error("HDF5: truncated structure");
    std::memcpy(&v, b.data() + at, sizeof(T));
    return v;
  }

  static Type datatype(const std::vector<uint8_t> &m, size_t at) {
    Type t;
    const uint8_t head = get<uint8_t>(m, at);
    t.cls = head & 0x0f;
    const int version = head >> 4;
    const uint8_t bits0 = get<uint8_t>(m, at + 1);
    t.size = get<uint32_t>(m, at + 4);
    if (t.cls == 0)
      t.is_signed = (bits0 & 0x08) != 0;
    if ((t.cls == 0 || t.cls == 1) && (bits0 & 0x01))
      throw std::runtime_error("HDF5: big-endian data are not read");
    if (t.cls == 6) {
      /* compound: class bits 0-15 = number of members; per member (versions
       * 1 and 2) its name padded to a multiple of 8 bytes, the byte offset,
       * (version 1: 28 bytes of array information,) its datatype message;
       * version 3: name not padded, offset in as many bytes as the size of
       * the compound needs */
      const unsigned count = bits0 | ((unsigned)get<uint8_t>(m, at + 2) << 8);
      size_t p = at + 8;
      for (unsigned k = 0; k < count; ++k) {
        Type::Member member;
        size_t end = p;
        while (get<uint8_t>(m, end) != 0)
          ++end;
        member.name.assign(reinterpret_cast<const char *>(m.data()) + p,
                           end - p);
        if (version < 3) {
          p += ((end - p) / 8 + 1) * 8;
          member.offset = get<uint32_t>(m, p);
          p += version == 1 ? 32 : 4;
        } else {
          p = end + 1;
          int nbytes = 1;
          while (nbytes < 4 && (t.size >> (8 * nbytes)) != 0)
            ++nbytes;
          for (int b = 0; b < nbytes; ++b)
            member.offset |= (uint32_t)get<uint8_t>(m, p + b) << (8 * b);
          p += nbytes;
        }
        const Type inner = datatype(m, p);
        if (inner.cls == 6)
          throw std::runtime_error("HDF5: nested compound types are not read");
        member.cls = inner.cls;
        member.size = inner.size;
        member.is_signed = inner.is_signed;
        p += inner.message_bytes;
        t.members.push_back(member);
      }
      t.message_bytes = p - at;
      return t;
    }
    if (t.cls != 0 && t.cls != 1 && t.cls != 3)
      throw std::runtime_error("HDF5: datatype class " +
                               std::to_string(t.cls) + " is not read");
    t.message_bytes = t.cls == 0 ? 12 : (t.cls == 1 ? 20 : 8);
    return t;
  }
  /* size of a datatype message (to find what follows it in an attribute) */
  static size_t datatype_bytes(const Type &t) { return t.message_bytes; }
  static std::vector<uint64_t> dataspace(const std::vector<uint8_t> &m,
                                         size_t at, size_t *used = nullptr) {
    const uint8_t version = get<uint8_t>(m, at);
    const uint8_t rank = get<uint8_t>(m, at + 1);
    const uint8_t flags = get<uint8_t>(m, at + 2);
    size_t p = at + (version == 1 ? 8 : 4);
    std::vector<uint64_t> dims(rank);
    for (int i = 0; i < rank; ++i, p += 8)
      dims[i] = get<uint64_t>(m, p);
    if (flags & 1)
      p += 8 * rank; /* maximum dimensions */
    if (used)
      *used = p - at;
    return dims;
  }

  /* the messages of a version-1 object header, continuation blocks included */
  void messages(uint64_t address,
                std::vector<std::pair<uint16_t, std::vector<uint8_t>>> &out)
      const {
    const std::vector<uint8_t> head = bytes(address, 16);
    if (head[0] == 'O' && head[1] == 'H' && head[2] == 'D' && head[3] == 'R')
      error("version-2 object headers (libver=latest) are not read");
    if (head[0] != 1)
      error("object header version " + std::to_string(head[0]));
    uint32_t left = get<uint16_t>(head, 2);
    std::vector<std::pair<uint64_t, uint64_t>> blocks;
    blocks.emplace_back(address + 16, (uint64_t)get<uint32_t>(head, 8));
    for (size_t k = 0; k < blocks.size() && left > 0; ++k) {
      const std::vector<uint8_t> b = bytes(blocks[k].first, blocks[k].second);
      size_t p = 0;
      while (left > 0 && p + 8 <= b.size()) {
        const uint16_t type = get<uint16_t>(b, p);
        const uint16_t size = get<uint16_t>(b, p + 2);
        if (p + 8 + size > b.size())
          error("object header message overruns its block");
        std::vector<uint8_t> data(b.begin() + p + 8, b.begin() + p + 8 + size);
        p += 8 + size;
        --left;
        if (type == 0x0010)
          blocks.emplace_back(get<uint64_t>(data, 0), get<uint64_t>(data, 8));
        else
          out.emplace_back(type, std::move(data));
      }
    }
  }

  std::string heap_string(const std::vector<uint8_t> &heap, uint64_t at) const {
    if (at >= heap.size())
      error("name outside the local heap");
    const char *s = reinterpret_cast<const char *>(heap.data() + at);
    return std::string(s, strnlen(s, heap.size() - at));
  }

  /* symbol-table entries below a version-1 group B-tree node */
  void group_node(uint64_t address, const std::vector<uint8_t> &heap,
                  std::map<std::string, uint64_t> &members, int depth) const {
    if (depth > 32)
      error("group B-tree too deep");
    const std::vector<uint8_t> head = bytes(address, 8);
    if (std::memcmp(head.data(), "SNOD", 4) == 0) {
      const uint16_t n = get<uint16_t>(head, 6);
      const std::vector<uint8_t> b = bytes(address + 8, 40ull * n);
      for (uint16_t i = 0; i < n; ++i)
        members[heap_string(heap, get<uint64_t>(b, 40 * i))] =
            get<uint64_t>(b, 40 * i + 8);
      return;
    }
    if (std::memcmp(head.data(), "TREE", 4) != 0 || head[4] != 0)
      error("not a group B-tree node");
    const uint16_t n = get<uint16_t>(head, 6);
    /* signature 4, type 1, level 1, entries 2, siblings 16; then key 0,
     * child 0, key 1, ... */
    const std::vector<uint8_t> b = bytes(address + 24, 16ull * n + 8);
    for (uint16_t i = 0; i < n; ++i)
      group_node(get<uint64_t>(b, 16 * i + 8), heap, members, depth + 1);
  }

  /* the chunks below a version-1 chunk B-tree node */
  struct Chunk {
    uint32_t bytes, mask;
    std::vector<uint64_t> offset;
    uint64_t address;
  };
  void chunk_node(uint64_t address, size_t rank, std::vector<Chunk> &chunks,
                  int depth) const {
    if (depth > 32)
      error("chunk B-tree too deep");
    const std::vector<uint8_t> head = bytes(address, 24);
    if (std::memcmp(head.data(), "TREE", 4) != 0 || head[4] != 1)
      error("not a chunk B-tree node");
    const uint8_t level = head[5];
    const uint16_t n = get<uint16_t>(head, 6);
    const size_t key = 8 + 8 * (rank + 1);
    const std::vector<uint8_t> b = bytes(address + 24, (key + 8) * n + key);
    for (uint16_t i = 0; i < n; ++i) {
      const size_t p = (key + 8) * i;
      const uint64_t child = get<uint64_t>(b, p + key);
      if (level > 0) {
        chunk_node(child, rank, chunks, depth + 1);
        continue;
      }
      Chunk c;
      c.bytes = get<uint32_t>(b, p);
      c.mask = get<uint32_t>(b, p + 4);
      c.offset.resize(rank);
      for (size_t d = 0; d < rank; ++d)
        c.offset[d] = get<uint64_t>(b, p + 8 + 8 * d);
      c.address = child;
      chunks.push_back(std::move(c));
    }
  }

  static void inflate_chunk(std::vector<uint8_t> &data, size_t expected) {
    std::vector<uint8_t> out(expected);
    uLongf n = (uLongf)expected;
    const int rc = uncompress(out.data(), &n, data.data(), (uLong)data.size());
    if (rc != Z_OK || n != expected)
      throw std::runtime_error("HDF5: a deflated chunk does not inflate to "
                               "its size");
    data.swap(out);
  }
  static void unshuffle(std::vector<uint8_t> &data, size_t element) {
    if (element < 2)
      return;
    const size_t n = data.size() / element;
    std::vector<uint8_t> out(data.size());
    for (size_t b = 0; b < element; ++b)
      for (size_t i = 0; i < n; ++i)
        out[i * element + b] = data[b * n + i];
    for (size_t i = n * element; i < data.size(); ++i)
      out[i] = data[i];
    data.swap(out);
  }

public:
  explicit Hdf5Reader(const std::string &filename)
      : _name(filename), _file(filename, std::ios::binary) {
    if (!_file)
      throw std::runtime_error("Could not open file \"" + filename + "\"!");
    _file.seekg(0, std::ios::end);
    _size = (uint64_t)_file.tellg();
    /* the superblock sits at 0, 512, 1024, ... */
    static const uint8_t signature[8] = {0x89, 'H', 'D', 'F',
                                         '\r', '\n', 0x1a, '\n'};
    uint64_t at = 0;
    for (;; at = at ? 2 * at : 512) {
      if (at + 8 > _size)
        error("no HDF5 signature");
      const std::vector<uint8_t> s = bytes(at, 8);
      if (std::memcmp(s.data(), signature, 8) == 0)
        break;
    }
    const std::vector<uint8_t> sb = bytes(at, 96 + 8);
    const uint8_t version = sb[8];
    if (version > 1)
      error("superblock version " + std::to_string(version) +
            " (libver=latest) is not read");
    if (sb[13] != 8 || sb[14] != 8)
      error("offsets and lengths must be 8 bytes wide");
    const size_t p = version == 0 ? 24 : 28;
    _base = get<uint64_t>(sb, p);
    /* root group symbol table entry: link name offset, header address */
    _root = get<uint64_t>(sb, p + 32 + 8);
    _base += 0; /* addresses are relative to the base address */
  }

  /* the object at a path such as "/PartType0/Temperature" */
  Object open(const std::string &path) const {
    uint64_t address = _root;
    size_t at = 0;
    while (at < path.size()) {
      while (at < path.size() && path[at] == '/')
        ++at;
      if (at >= path.size())
        break;
      const size_t end = path.find('/', at);
      const std::string part = path.substr(at, end - at);
      const Object group = object(address);
      const auto it = group.members.find(part);
      if (it == group.members.end())
        error("no object \"" + path + "\"");
      address = it->second;
      at = end == std::string::npos ? path.size() : end;
    }
    return object(address);
  }

  bool exists(const std::string &path) const {
    try {
      (void)open(path);
      return true;
    } catch (const std::runtime_error &) {
      return false;
    }
  }

  Object object(uint64_t address) const {
    std::vector<std::pair<uint16_t, std::vector<uint8_t>>> msgs;
    messages(address, msgs);
    Object o;
    for (const auto &m : msgs) {
      const std::vector<uint8_t> &d = m.second;
      switch (m.first) {
      case 0x0011: { /* symbol table: B-tree, local heap */
        o.is_group = true;
        const uint64_t btree = get<uint64_t>(d, 0);
        const uint64_t heap = get<uint64_t>(d, 8);
        const std::vector<uint8_t> hh = bytes(heap, 32);
        if (std::memcmp(hh.data(), "HEAP", 4) != 0)
          error("not a local heap");
        const std::vector<uint8_t> segment =
            bytes(get<uint64_t>(hh, 24), get<uint64_t>(hh, 8));
        group_node(btree, segment, o.members, 0);
        break;
      }
      case 0x0001:
        o.dims = dataspace(d, 0);
        break;
      case 0x0003:
        o.type = datatype(d, 0);
        break;
      case 0x0008: {
        if (d[0] != 3)
          error("data layout message version " + std::to_string(d[0]));
        o.layout = d[1];
        if (o.layout == 0) {
          const uint16_t n = get<uint16_t>(d, 2);
          o.compact.assign(d.begin() + 4, d.begin() + 4 + n);
        } else if (o.layout == 1) {
          o.data_address = get<uint64_t>(d, 2);
          o.data_size = get<uint64_t>(d, 10);
        } else if (o.layout == 2) {
          const uint8_t rank = d[2]; /* dataset rank + 1 */
          o.data_address = get<uint64_t>(d, 3);
          o.chunk_dims.resize(rank);
          for (int i = 0; i < rank; ++i)
            o.chunk_dims[i] = get<uint32_t>(d, 11 + 4 * i);
        } else {
          error("data layout class " + std::to_string(o.layout));
        }
        break;
      }
      case 0x000b: { /* filter pipeline */
        const uint8_t version = d[0], n = d[1];
        size_t p = version == 1 ? 8 : 2;
        for (int i = 0; i < n; ++i) {
          const uint16_t id = get<uint16_t>(d, p);
          uint16_t name_length = 0;
          if (version == 1 || id >= 256) {
            name_length = get<uint16_t>(d, p + 2);
            p += 2;
          }
          const uint16_t nvalues = get<uint16_t>(d, p + 4);
          p += 6;
          if (version == 1)
            name_length = (uint16_t)((name_length + 7) & ~7);
          p += name_length + 4 * nvalues;
          if (version == 1 && (nvalues & 1))
            p += 4;
          o.filters.push_back(id);
        }
        break;
      }
      case 0x000c: { /* attribute */
        const uint8_t version = d[0];
        if (version < 1 || version > 3)
          error("attribute message version " + std::to_string(version));
        const uint16_t name_size = get<uint16_t>(d, 2);
        const uint16_t type_size = get<uint16_t>(d, 4);
        const uint16_t space_size = get<uint16_t>(d, 6);
        size_t p = version == 3 ? 9 : 8;
        auto padded = [version](size_t n) {
          return version == 1 ? (n + 7) & ~(size_t)7 : n;
        };
        const std::string name(reinterpret_cast<const char *>(d.data() + p),
                               strnlen(reinterpret_cast<const char *>(
                                           d.data() + p),
                                       name_size));
        p += padded(name_size);
        Attribute a;
        a.type = datatype(d, p);
        p += padded(type_size);
        a.dims = dataspace(d, p);
        p += padded(space_size);
        uint64_t count = 1;
        for (uint64_t n : a.dims)
          count *= n;
        if (p + count * a.type.size > d.size())
          error("attribute \"" + name + "\" overruns its message");
        a.data.assign(d.begin() + p, d.begin() + p + count * a.type.size);
        o.attributes[name] = std::move(a);
        break;
      }
      default:
        break;
      }
    }
    return o;
  }

  /* HDF5Tools::read_attribute< std::string > */
  static std::string as_string(const Attribute &a) {
    if (a.type.cls != 3)
      throw std::runtime_error("HDF5: attribute is not a string");
    const char *s = reinterpret_cast<const char *>(a.data.data());
    return std::string(s, strnlen(s, a.data.size()));
  }
  /* numbers of any of the supported types, as doubles */
  static std::vector<double> as_doubles(const Type &t, const uint8_t *data,
                                        uint64_t count) {
    std::vector<double> out(count);
    for (uint64_t i = 0; i < count; ++i) {
      const uint8_t *p = data + i * t.size;
      if (t.cls == 1 && t.size == 8) {
        std::memcpy(&out[i], p, 8);
      } else if (t.cls == 1 && t.size == 4) {
        float f;
        std::memcpy(&f, p, 4);
        out[i] = f;
      } else if (t.cls == 0 && t.size <= 8) {
        uint64_t u = 0;
        std::memcpy(&u, p, t.size);
        if (t.is_signed && t.size < 8 && (u >> (8 * t.size - 1)))
          u |= ~(uint64_t)0 << (8 * t.size);
        out[i] = t.is_signed ? (double)(int64_t)u : (double)u;
      } else {
        throw std::runtime_error("HDF5: not a number type");
      }
    }
    return out;
  }
  static std::vector<double> as_doubles(const Attribute &a) {
    return as_doubles(a.type, a.data.data(), a.data.size() / a.type.size);
  }

  /* the raw bytes of a dataset, in row-major order */
  std::vector<uint8_t> raw(const Object &o) const {
    if (o.layout < 0 || o.type.cls < 0)
      error("not a dataset");
    uint64_t count = 1;
    for (uint64_t n : o.dims)
      count *= n;
    const uint64_t total = count * o.type.size;
    if (o.layout == 0) {
      if (o.compact.size() < total)
        error("compact dataset shorter than its dataspace");
      return std::vector<uint8_t>(o.compact.begin(), o.compact.begin() + total);
    }
    if (o.layout == 1) {
      if (o.data_address == UNDEF)
        return std::vector<uint8_t>(total, 0); /* never written */
      if (o.data_size < total)
        error("contiguous dataset shorter than its dataspace");
      return bytes(o.data_address, total);
    }
    /* chunked */
    const size_t rank = o.dims.size();
    if (rank < 1 || o.chunk_dims.size() != rank + 1)
      error("chunk rank does not match the dataspace");
    for (uint16_t f : o.filters)
      if (f != 1 && f != 2)
        error("filter " + std::to_string(f) + " is not read (deflate and "
              "shuffle are)");
    std::vector<uint8_t> out(total, 0);
    if (o.data_address == UNDEF)
      return out;
    std::vector<Chunk> chunks;
    chunk_node(o.data_address, rank, chunks, 0);
    uint64_t chunk_elements = 1;
    for (size_t d = 0; d < rank; ++d)
      chunk_elements *= o.chunk_dims[d];
    const size_t element = o.type.size;
    const uint64_t chunk_bytes = chunk_elements * element;
    for (const Chunk &c : chunks) {
      std::vector<uint8_t> data = bytes(c.address, c.bytes);
      /* filters are undone last to first; mask bit i = filter i skipped */
      for (size_t k = o.filters.size(); k-- > 0;) {
        if (c.mask & (1u << k))
          continue;
        if (o.filters[k] == 1)
          inflate_chunk(data, chunk_bytes);
        else
          unshuffle(data, element);
      }
      if (data.size() < chunk_bytes)
        error("a chunk is shorter than its dimensions");
      /* copy the part of the chunk that lies inside the dataset */
      /* (row by row along the last dimension) */
      std::vector<uint64_t> idx(rank, 0);
      const uint64_t row = o.chunk_dims[rank - 1];
      const uint64_t first = c.offset[rank - 1];
      const uint64_t last_dim = o.dims[rank - 1];
      if (first >= last_dim)
        continue;
      for (uint64_t e = 0; e < chunk_elements; e += row) {
        uint64_t rest = e / row;
        for (size_t d = rank - 1; d-- > 0;) {
          idx[d] = rest % o.chunk_dims[d];
          rest /= o.chunk_dims[d];
        }
        bool inside = true;
        uint64_t dest = 0;
        for (size_t d = 0; d + 1 < rank; ++d) {
          const uint64_t g = c.offset[d] + idx[d];
          if (g >= o.dims[d])
            inside = false;
          dest = dest * o.dims[d] + g;
        }
        if (!inside)
          continue;
        const uint64_t n = std::min<uint64_t>(row, last_dim - first);
        std::memcpy(out.data() + (dest * last_dim + first) * element,
                    data.data() + e * element, n * element);
      }
    }
    return out;
  }

  /* HDF5Tools::read_dictionary (src/HDF5Tools.hpp:1228-1330): a dataset of
   * {name: fixed-length string, value: number} records as a map; trailing
   * spaces of the names stripped */
  std::map<std::string, double> read_dictionary(const std::string &path) const {
    const Object o = open(path);
    if (o.type.cls != 6)
      error("\"" + path + "\" is not a dataset of {name, value} records");
    const Type::Member *name = nullptr, *value = nullptr;
    for (const Type::Member &member : o.type.members) {
      if (member.name == "name" && member.cls == 3)
        name = &member;
      if (member.name == "value" && (member.cls == 0 || member.cls == 1))
        value = &member;
    }
    if (!name || !value)
      error("\"" + path + "\" has no {name, value} records");
    if ((uint64_t)name->offset + name->size > o.type.size ||
        (uint64_t)value->offset + value->size > o.type.size)
      error("\"" + path + "\": a member lies outside its record");
    const std::vector<uint8_t> b = raw(o);
    Type number;
    number.cls = value->cls;
    number.size = value->size;
    number.is_signed = value->is_signed;
    std::map<std::string, double> dictionary;
    for (size_t at = 0; at + o.type.size <= b.size(); at += o.type.size) {
      const char *text = reinterpret_cast<const char *>(b.data()) + at +
                         name->offset;
      size_t length = strnlen(text, name->size);
      while (length > 0 && text[length - 1] == ' ')
        --length;
      dictionary[std::string(text, length)] =
          as_doubles(number, b.data() + at + value->offset, 1)[0];
    }
    return dictionary;
  }

  /* HDF5Tools::read_dataset< double >: any number type, as doubles */
  std::vector<double> read_doubles(const std::string &path) const {
    const Object o = open(path);
    const std::vector<uint8_t> b = raw(o);
    return as_doubles(o.type, b.data(), b.size() / o.type.size);
  }
};

} // namespace cmi

#endif
