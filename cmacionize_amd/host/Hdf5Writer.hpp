/*
 * Hdf5Writer.hpp - a minimal, dependency-free writer of HDF5 files: just what
 * the reference's snapshot format needs (GadgetDensityGridWriter,
 * src/GadgetDensityGridWriter.cpp:107-358, through src/HDF5Tools.hpp): one
 * level of groups under the root, contiguous little-endian datasets of
 * doubles (1-D or [n][3]), and group attributes that are scalars or small 1-D
 * arrays of double / int32 / uint32 or fixed-length strings.
 *
 * The image has no HDF5 development files, so the file is laid out by hand
 * after the HDF5 File Format Specification version 1.1 ("HDF5 1.6" layout,
 * readable by every HDF5 library and h5py): superblock version 0, version-1
 * object headers, groups as symbol tables (one version-1 B-tree node with one
 * symbol-table node each, names in a local heap), attributes as header
 * messages, data layout version 3 (contiguous). All metadata sits at the
 * start of the file, the raw data of the datasets follows in order and is
 * streamed from the caller's arrays.
 */
#ifndef CMI_HDF5WRITER_HPP
#define CMI_HDF5WRITER_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace cmi {

class Hdf5Writer {
  /* entries a symbol-table node can hold (2 x the group leaf node K of the
   * superblock): one node per group is enough for every group written here */
  static constexpr int LEAF_K = 128;
  static constexpr int INTERNAL_K = 16;
  static constexpr uint64_t UNDEF = ~(uint64_t)0;

  struct Bytes : std::vector<uint8_t> {
    void u8(uint8_t v) { push_back(v); }
    void u16(uint16_t v) { raw(&v, 2); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void raw(const void *p, size_t n) {
      const uint8_t *b = static_cast<const uint8_t *>(p);
      insert(end(), b, b + n);
    }
    void zeros(size_t n) { insert(end(), n, (uint8_t)0); }
    void pad8() { zeros((8 - size() % 8) % 8); }
  };

  /* datatype messages (section IV.A.2.d of the specification) */
  static Bytes type_double() {
    Bytes t;
    t.u8(0x11);                 /* version 1, class 1 (floating point) */
    t.u8(0x20);                 /* little-endian, implied mantissa bit */
    t.u8(63);                   /* sign bit position */
    t.u8(0);
    t.u32(8);                   /* size */
    t.u16(0);                   /* bit offset */
    t.u16(64);                  /* precision */
    t.u8(52);                   /* exponent location */
    t.u8(11);                   /* exponent size */
    t.u8(0);                    /* mantissa location */
    t.u8(52);                   /* mantissa size */
    t.u32(1023);                /* exponent bias */
    return t;
  }
  static Bytes type_int32(bool is_signed) {
    Bytes t;
    t.u8(0x10); /* version 1, class 0 (fixed point) */
    t.u8(is_signed ? 0x08 : 0x00);
    t.u8(0);
    t.u8(0);
    t.u32(4);
    t.u16(0);  /* bit offset */
    t.u16(32); /* precision */
    return t;
  }
  static Bytes type_string(uint32_t size) {
    Bytes t;
    t.u8(0x13); /* version 1, class 3 (string) */
    t.u8(0x00); /* null terminated, ASCII */
    t.u8(0);
    t.u8(0);
    t.u32(size);
    return t;
  }
  /* dataspace message, version 1 */
  static Bytes space(const std::vector<uint64_t> &dims) {
    Bytes s;
    s.u8(1);
    s.u8((uint8_t)dims.size());
    s.u8(0);
    s.zeros(5);
    for (uint64_t d : dims)
      s.u64(d);
    return s;
  }

  struct Message {
    uint16_t type;
    Bytes data;
  };
  struct Dataset {
    std::string name;
    std::vector<uint64_t> dims;
    /* writes the raw data (dims product elements) to the stream */
    std::function<void(std::ostream &)> write;
    uint64_t bytes;
    Bytes type; /* datatype message of an element; empty: IEEE double */
  };
  struct Group {
    std::string name;
    std::vector<Message> attributes;
    std::vector<Dataset> datasets;
  };
  std::vector<Group> _groups;

  static Message attribute(const std::string &name, const Bytes &type,
                           const std::vector<uint64_t> &dims, const void *data,
                           size_t bytes) {
    const Bytes sp = space(dims);
    Message m;
    m.type = 0x000C;
    m.data.u8(1); /* version */
    m.data.u8(0);
    m.data.u16((uint16_t)(name.size() + 1));
    m.data.u16((uint16_t)type.size());
    m.data.u16((uint16_t)sp.size());
    m.data.raw(name.c_str(), name.size() + 1);
    m.data.pad8();
    m.data.raw(type.data(), type.size());
    m.data.pad8();
    m.data.raw(sp.data(), sp.size());
    m.data.pad8();
    m.data.raw(data, bytes);
    if (m.data.size() > 65000)
      throw std::runtime_error("HDF5 attribute \"" + name + "\" too large");
    return m;
  }

  Group &group(const std::string &name) {
    for (Group &g : _groups)
      if (g.name == name)
        return g;
    Group g;
    g.name = name;
    _groups.push_back(g);
    return _groups.back();
  }

  /* version-1 object header holding the messages */
  static Bytes object_header(const std::vector<Message> &messages) {
    Bytes body;
    for (const Message &m : messages) {
      Bytes d = m.data;
      d.pad8();
      body.u16(m.type);
      body.u16((uint16_t)d.size());
      body.u8(0);
      body.zeros(3);
      body.raw(d.data(), d.size());
    }
    Bytes h;
    h.u8(1);
    h.u8(0);
    h.u16((uint16_t)messages.size());
    h.u32(1); /* reference count */
    h.u32((uint32_t)body.size());
    h.zeros(4); /* messages start on an 8-byte boundary */
    h.raw(body.data(), body.size());
    return h;
  }

  /* size of the structures of a group with the given member names */
  static uint64_t heap_data_size(const std::vector<std::string> &names) {
    uint64_t n = 8; /* the empty name at offset 0 */
    for (const std::string &s : names)
      n += (s.size() + 1 + 7) / 8 * 8;
    return n;
  }
  static constexpr uint64_t BTREE_BYTES =
      24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8;
  static constexpr uint64_t SNOD_BYTES = 8 + LEAF_K * 40;
  static constexpr uint64_t HEAP_HEADER_BYTES = 32;

  /* B-tree node + symbol-table node + local heap of a group, at `at`;
   * members = (name, object header address), any order */
  static Bytes group_tables(uint64_t at,
                            std::vector<std::pair<std::string, uint64_t>> members,
                            uint64_t &btree, uint64_t &heap) {
    if ((int)members.size() > LEAF_K)
      throw std::runtime_error("HDF5 group with too many members");
    std::sort(members.begin(), members.end());
    std::vector<std::string> names;
    for (auto &m : members)
      names.push_back(m.first);
    btree = at;
    const uint64_t snod = btree + BTREE_BYTES;
    heap = snod + SNOD_BYTES;
    const uint64_t heap_data = heap + HEAP_HEADER_BYTES;
    /* name offsets in the heap */
    std::vector<uint64_t> offset;
    uint64_t pos = 8;
    for (const std::string &s : names) {
      offset.push_back(pos);
      pos += (s.size() + 1 + 7) / 8 * 8;
    }
    Bytes b;
    b.raw("TREE", 4);
    b.u8(0); /* group node */
    b.u8(0); /* leaf level */
    b.u16(members.empty() ? 0 : 1);
    b.u64(UNDEF);
    b.u64(UNDEF);
    b.u64(0); /* key 0: the empty name */
    b.u64(members.empty() ? UNDEF : snod);
    b.u64(members.empty() ? 0 : offset.back()); /* key 1: largest name */
    b.zeros(btree + BTREE_BYTES - at - b.size());
    b.raw("SNOD", 4);
    b.u8(1);
    b.u8(0);
    b.u16((uint16_t)members.size());
    for (size_t i = 0; i < members.size(); ++i) {
      b.u64(offset[i]);
      b.u64(members[i].second);
      b.u32(0); /* nothing cached */
      b.u32(0);
      b.zeros(16);
    }
    b.zeros((LEAF_K - members.size()) * 40);
    b.raw("HEAP", 4);
    b.u8(0);
    b.zeros(3);
    b.u64(heap_data_size(names));
    b.u64(1); /* head of the free list: H5HL_FREE_NULL, no free block */
    b.u64(heap_data);
    b.zeros(8);
    for (const std::string &s : names) {
      b.raw(s.c_str(), s.size() + 1);
      b.pad8();
    }
    return b;
  }

public:
  /* attributes of a group (created on first use, in call order) */
  void attribute(const std::string &group_name, const std::string &name,
                 double value) {
    group(group_name).attributes.push_back(
        attribute(name, type_double(), {}, &value, 8));
  }
  void attribute(const std::string &group_name, const std::string &name,
                 int32_t value) {
    group(group_name).attributes.push_back(
        attribute(name, type_int32(true), {}, &value, 4));
  }
  void attribute(const std::string &group_name, const std::string &name,
                 uint32_t value) {
    group(group_name).attributes.push_back(
        attribute(name, type_int32(false), {}, &value, 4));
  }
  void attribute(const std::string &group_name, const std::string &name,
                 const std::vector<double> &value) {
    group(group_name).attributes.push_back(attribute(
        name, type_double(), {value.size()}, value.data(), 8 * value.size()));
  }
  void attribute(const std::string &group_name, const std::string &name,
                 const std::vector<uint32_t> &value) {
    group(group_name).attributes.push_back(
        attribute(name, type_int32(false), {value.size()}, value.data(),
                  4 * value.size()));
  }
  void attribute(const std::string &group_name, const std::string &name,
                 const std::string &value) {
    group(group_name).attributes.push_back(
        attribute(name, type_string((uint32_t)value.size() + 1), {},
                  value.c_str(), value.size() + 1));
  }
  void create_group(const std::string &group_name) { (void)group(group_name); }
  /* an attribute as another file held it: class 0 (integer), 1 (IEEE float)
   * or 3 (fixed-length string), element size, dimensions, raw bytes */
  void attribute_raw(const std::string &group_name, const std::string &name,
                     int cls, uint32_t size, bool is_signed,
                     const std::vector<uint64_t> &dims,
                     const std::vector<uint8_t> &data) {
    Bytes type;
    if (cls == 1 && size == 8)
      type = type_double();
    else if (cls == 0 && size == 4)
      type = type_int32(is_signed);
    else if (cls == 3)
      type = type_string(size);
    else
      throw std::runtime_error("HDF5 attribute \"" + name +
                               "\": a type this writer does not write");
    group(group_name).attributes.push_back(
        attribute(name, type, dims, data.data(), data.size()));
  }

  /* a dataset of doubles; `write` streams prod(dims) doubles when the file is
   * written */
  void dataset(const std::string &group_name, const std::string &name,
               const std::vector<uint64_t> &dims,
               std::function<void(std::ostream &)> write) {
    Dataset d;
    d.name = name;
    d.dims = dims;
    d.write = write;
    d.bytes = 8;
    for (uint64_t n : dims)
      d.bytes *= n;
    group(group_name).datasets.push_back(d);
  }
  void dataset(const std::string &group_name, const std::string &name,
               const std::vector<double> &values) {
    const std::vector<double> *v = &values;
    dataset(group_name, name, {values.size()}, [v](std::ostream &os) {
      os.write(reinterpret_cast<const char *>(v->data()), 8 * v->size());
    });
  }

  /* a 1-D dataset of fixed-length, null-padded strings (the reference writes
   * variable-length ones, HDF5Tools::write_dataset for std::string: readers
   * see the same strings) */
  void dataset(const std::string &group_name, const std::string &name,
               const std::vector<std::string> &values) {
    size_t width = 1;
    for (const std::string &v : values)
      width = std::max(width, v.size() + 1);
    std::string raw(width * values.size(), '\0');
    for (size_t i = 0; i < values.size(); ++i)
      raw.replace(i * width, values[i].size(), values[i]);
    Dataset d;
    d.name = name;
    d.dims = {values.size()};
    d.type = type_string((uint32_t)width);
    d.bytes = raw.size();
    d.write = [raw](std::ostream &os) { os.write(raw.data(), raw.size()); };
    group(group_name).datasets.push_back(d);
  }

  void write(const std::string &filename) const {
    /* pass 1: sizes -> addresses. Layout: superblock, root header, root
     * tables, then per group {header, tables, dataset headers}, raw data. */
    const uint64_t superblock_bytes = 96;
    auto dataset_header = [](const Dataset &d, uint64_t address) {
      std::vector<Message> m(4);
      m[0].type = 0x0001;
      m[0].data = space(d.dims);
      m[1].type = 0x0003;
      m[1].data = d.type.empty() ? type_double() : d.type;
      m[2].type = 0x0005; /* fill value, version 2: none defined */
      m[2].data.u8(2);
      m[2].data.u8(2); /* space allocation time: late (1 is early) - what
                        * libhdf5 itself writes for contiguous data */
      m[2].data.u8(0); /* fill value written on allocation */
      m[2].data.u8(0); /* no fill value defined */
      m[3].type = 0x0008; /* layout version 3, contiguous */
      m[3].data.u8(3);
      m[3].data.u8(1);
      m[3].data.u64(address);
      m[3].data.u64(d.bytes);
      return object_header(m);
    };
    auto group_header = [](const Group &g, uint64_t btree, uint64_t heap) {
      std::vector<Message> m;
      Message st;
      st.type = 0x0011;
      st.data.u64(btree);
      st.data.u64(heap);
      m.push_back(st);
      m.insert(m.end(), g.attributes.begin(), g.attributes.end());
      return object_header(m);
    };
    auto tables_bytes = [](const std::vector<std::string> &names) {
      return BTREE_BYTES + SNOD_BYTES + HEAP_HEADER_BYTES +
             heap_data_size(names);
    };
    Group root;
    uint64_t at = superblock_bytes;
    const uint64_t root_header_at = at;
    at += group_header(root, 0, 0).size();
    std::vector<std::string> group_names;
    for (const Group &g : _groups)
      group_names.push_back(g.name);
    const uint64_t root_tables_at = at;
    at += tables_bytes(group_names);
    struct Placement {
      uint64_t header, tables;
      std::vector<uint64_t> dataset_header;
    };
    std::vector<Placement> place(_groups.size());
    for (size_t k = 0; k < _groups.size(); ++k) {
      const Group &g = _groups[k];
      place[k].header = at;
      at += group_header(g, 0, 0).size();
      std::vector<std::string> names;
      for (const Dataset &d : g.datasets)
        names.push_back(d.name);
      place[k].tables = at;
      at += tables_bytes(names);
      for (const Dataset &d : g.datasets) {
        place[k].dataset_header.push_back(at);
        at += dataset_header(d, 0).size();
      }
    }
    at = (at + 7) / 8 * 8;
    const uint64_t data_at = at;
    std::vector<std::vector<uint64_t>> data_address(_groups.size());
    for (size_t k = 0; k < _groups.size(); ++k)
      for (const Dataset &d : _groups[k].datasets) {
        data_address[k].push_back(at);
        at += d.bytes;
      }
    const uint64_t end_of_file = at;

    /* pass 2: the metadata block */
    Bytes f;
    uint64_t root_btree = 0, root_heap = 0;
    std::vector<std::pair<std::string, uint64_t>> members;
    for (size_t k = 0; k < _groups.size(); ++k)
      members.push_back({_groups[k].name, place[k].header});
    const Bytes root_tables =
        group_tables(root_tables_at, members, root_btree, root_heap);
    static const uint8_t signature[8] = {0x89, 'H', 'D', 'F',
                                         '\r', '\n', 0x1a, '\n'};
    f.raw(signature, 8);
    f.u8(0); /* superblock version */
    f.u8(0); /* free-space storage version */
    f.u8(0); /* root group symbol table entry version */
    f.u8(0);
    f.u8(0); /* shared header message format version */
    f.u8(8); /* size of offsets */
    f.u8(8); /* size of lengths */
    f.u8(0);
    f.u16(LEAF_K / 2);  /* group leaf node K */
    f.u16(INTERNAL_K);  /* group internal node K */
    f.u32(0);           /* file consistency flags */
    f.u64(0);           /* base address */
    f.u64(UNDEF);       /* free-space information */
    f.u64(end_of_file);
    f.u64(UNDEF);       /* driver information block */
    f.u64(0);           /* root entry: link name offset */
    f.u64(root_header_at);
    f.u32(1);           /* cached: B-tree and heap of the group */
    f.u32(0);
    f.u64(root_btree);
    f.u64(root_heap);
    if (f.size() != superblock_bytes)
      throw std::logic_error("HDF5 superblock size");
    const Bytes rh = group_header(root, root_btree, root_heap);
    f.raw(rh.data(), rh.size());
    f.raw(root_tables.data(), root_tables.size());
    for (size_t k = 0; k < _groups.size(); ++k) {
      const Group &g = _groups[k];
      if (f.size() != place[k].header)
        throw std::logic_error("HDF5 layout (group header)");
      members.clear();
      for (size_t i = 0; i < g.datasets.size(); ++i)
        members.push_back({g.datasets[i].name, place[k].dataset_header[i]});
      uint64_t btree = 0, heap = 0;
      const Bytes tables = group_tables(place[k].tables, members, btree, heap);
      const Bytes gh = group_header(g, btree, heap);
      f.raw(gh.data(), gh.size());
      f.raw(tables.data(), tables.size());
      for (size_t i = 0; i < g.datasets.size(); ++i) {
        if (f.size() != place[k].dataset_header[i])
          throw std::logic_error("HDF5 layout (dataset header)");
        const Bytes dh = dataset_header(g.datasets[i], data_address[k][i]);
        f.raw(dh.data(), dh.size());
      }
    }
    f.zeros(data_at - f.size());

    std::ofstream os(filename, std::ios::binary | std::ios::trunc);
    if (!os)
      throw std::runtime_error("Unable to open file \"" + filename +
                               "\" for writing");
    os.write(reinterpret_cast<const char *>(f.data()), f.size());
    for (size_t k = 0; k < _groups.size(); ++k)
      for (const Dataset &d : _groups[k].datasets) {
        const std::streampos before = os.tellp();
        d.write(os);
        if ((uint64_t)(os.tellp() - before) != d.bytes)
          throw std::logic_error("HDF5 dataset \"" + d.name +
                                 "\": wrong number of bytes written");
      }
    if (!os)
      throw std::runtime_error("Error while writing \"" + filename + "\"");
  }
};

} // namespace cmi

#endif
