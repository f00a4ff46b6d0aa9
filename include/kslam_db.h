/*
 * kslam_db.h -- C ABI of the database load in front of kslam_set_index and the
 * host tail.  Same library as kslam.h; host-only.
 *
 * Replaces, in the reference (citations into /root/reference/):
 *
 *   getIndexFromBoostSerial          src/GenbankTools.h:336-344   (reader)
 *   GenbankIndex::writeIndexToBoostSerial  src/GenbankTools.h:201-205 (writer)
 *   the serialize() members          src/GenbankTools.h:58-62 (CDS), 101-109 (Gene),
 *                                    155-163 (GenbankEntry), 198-200 (GenbankIndex)
 *
 * <db>/database is a boost::archive::text_oarchive of a GenbankIndex: one line
 * of space-separated tokens (a 5 Gb database is one ~5 GB line).  The reference
 * reads it through Boost.Serialization into a vector of objects holding
 * std::strings; here the caller hands over the file's bytes (mmap'ed or read)
 * and gets the database by columns -- the kslam_index_view the host tail takes
 * as it is, and (pointer, length) pairs for kslam_set_index -- from two passes:
 * a token walk that only touches the numbers (a string is skipped by its
 * length prefix, so the walk over 5 GB reads a few megabytes), then a parallel
 * copy of the strings into their columns.
 *
 * PARITY UNPINNED.  Boost is not in the build image and the reference ships no
 * sample database, so neither the reference reader nor a real file could be
 * run against this parser.  The token grammar is the published behaviour of
 * Boost.Serialization text archives (boost/archive/basic_text_oprimitive.hpp,
 * detail/oserializer.hpp, serialization/collections_save_imp.hpp):
 *
 *   "22 serialization::archive" <library version>
 *   a class-type object, the FIRST time its type is saved: <tracking> <version>
 *       (both 0 here: nothing is saved through a pointer, no class is versioned)
 *   std::vector<T>:  <count> <item_version>, then the items
 *   std::string:     <length> ' ' <length raw bytes>
 *   bool: 0 | 1; integers in decimal; every token preceded by one space
 *
 * saved in the order  GenbankIndex{entries}; GenbankEntry{bases, taxonomyID,
 * genbankID, isPlasmid, is16S, locusTag, genes}; Gene{geneName, locusTag,
 * proteinID, product, referenceSequence, geneID, codingSequence}; CDS{start,
 * stop, complement}.  Types meet for the first time in the order GenbankIndex,
 * vector<GenbankEntry>, GenbankEntry, vector<Gene>, Gene, CDS.  The reader
 * accepts any library version and any whitespace between tokens, and -- since
 * the grammar could not be checked against a real file -- also the variants
 * without the class-info pair on the vector types and without <item_version>,
 * taking the first variant under which the whole text parses and every
 * count, length and flag is plausible; kslam_db_variant() says which one it was.
 */
#ifndef KSLAM_DB_H_
#define KSLAM_DB_H_
#include "kslam_tail.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kslam_db kslam_db;

/* Everything the archive holds, by columns.  `index` is the view the host
 * tail reads (kslam_tail.h); the rest are the serialised fields it does not
 * need, kept so that a database can be written back unchanged.  Strings of
 * column c: text[c_off[i] .. c_off[i + 1]). */
typedef struct {
  kslam_index_view index;         /* bases, locusTag, taxonomyID; genes: CDS start/stop, geneName, proteinID, product */
  const uint32_t *genbank_id;     /* n_entries */
  const uint8_t *is_plasmid;      /* n_entries, 0 | 1 */
  const uint8_t *is_16s;          /* n_entries, 0 | 1 */
  const char *gene_locus_tag;     /* Gene::locusTag */
  const uint64_t *gene_locus_tag_off;
  const char *gene_reference;     /* Gene::referenceSequence */
  const uint64_t *gene_reference_off;
  const uint32_t *gene_id;        /* n_genes */
  const uint8_t *gene_complement; /* CDS::complement, n_genes */
} kslam_db_columns;

/* Parses text[0 .. len) (the whole file).  threads: 0 = all usable CPUs.  On
 * error the message (with the byte offset where the walk stopped) is in
 * kslam_tail_last_error().  The columns are copies: `text` may be unmapped
 * after the call. */
kslam_status kslam_db_parse(const char *text, uint64_t len, int threads, kslam_db **out);
/* Same, from a file (mmap'ed for the duration of the call). */
kslam_status kslam_db_load(const char *path, int threads, kslam_db **out);
void kslam_db_free(kslam_db *db);

const kslam_db_columns *kslam_db_view(const kslam_db *db);
/* Boost library version found in the header. */
uint32_t kslam_db_library_version(const kslam_db *db);
/* Grammar variant the text parsed under: bit 0 set = no class-info pair on the
 * vector types, bit 1 set = no <item_version> after a vector's count.  0 is the
 * grammar in the header comment. */
uint32_t kslam_db_variant(const kslam_db *db);
/* Per-entry base pointers and lengths, the arguments of kslam_set_index
 * (kslam.h); valid until kslam_db_free. */
const char *const *kslam_db_entry_bases(const kslam_db *db);
const uint64_t *kslam_db_entry_lengths(const kslam_db *db);

/* Writes the columns as the reference's writeIndexToBoostSerial would (variant
 * 0 of the grammar, the given library version in the header, e.g. 17).  The
 * extra columns of `c` may be NULL (zeros / empty strings are written). */
kslam_status kslam_db_write(const char *path, const kslam_db_columns *c, uint32_t library_version);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_DB_H_ */
