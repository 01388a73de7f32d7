/* oracle/orc_internal.h -- internal types of the CPU oracle (test infrastructure, see orc.h). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------- Huffman / Codebook (Huffman.cs, Codebook.cs, Contracts/HuffmanListNode.cs) ---------- */
typedef struct {
  int value, length, bits, mask;
  int present; /* 0 == null node */
} orc_huff_node;

typedef struct {
  int dimensions, entries, map_type;
  int *lengths;
  float *lookup; /* entries*dimensions, NULL for map type 0 */
  orc_huff_node *prefix; /* 1<<prefix_bits slots (NULL if the tree was never built) */
  int prefix_count;
  orc_huff_node *overflow; /* NULL == the C# null list */
  int overflow_count;
  int prefix_bits, max_bits;
} orc_codebook;

int orc_codebook_init(orc_codebook *cb, orc_packet *p);
void orc_codebook_free(orc_codebook *cb);
/* Codebook.cs:294-320; returns -1 on failure, -2 on a C# NullReference-class fault */
int orc_decode_scalar(const orc_codebook *cb, orc_packet *p);

/* ---------- Floors ---------- */
typedef struct {
  /* Floor1.cs:21-25 */
  int partition_count;
  int partition_class[32];
  int class_count;
  int class_dimensions[16], class_subclasses[16], class_masterbook[16];
  int subclass_book[16][8]; /* book index or -1 */
  int multiplier, range, y_bits;
  int x_count;
  int x_list[256], l_neigh[256], h_neigh[256], sort_idx[256];
} orc_floor1;

typedef struct {
  /* Floor0.cs:22-26 */
  int order, rate, bark_map_size, amp_bits, amp_ofs, amp_div;
  int book_count, book_bits;
  int books[16];
  int *bark_map[2]; /* [0]=block0, [1]=block1; n/2+1 ints */
  float *w_map[2];  /* n/2 floats */
  int block_size[2];
} orc_floor0;

typedef struct {
  int type; /* 0 or 1 */
  orc_floor0 f0;
  orc_floor1 f1;
} orc_floor;

/* IFloorData (Contracts/IFloorData.cs, Floor1.cs:10-19, Floor0.cs:11-20) */
typedef struct {
  int type;
  int posts[256];
  int post_count;
  float coeff[257];
  float amp;
  int force_energy, force_no_energy;
} orc_floor_data;

int orc_floor_init(orc_floor *f, int type, orc_packet *p, int channels, int block0, int block1,
                   const orc_codebook *books, int nbooks);
void orc_floor_free(orc_floor *f);
int orc_floor_unpack(const orc_floor *f, const orc_codebook *books, orc_packet *p, int block_size,
                     orc_floor_data *out);
int orc_floor_execute_channel(const orc_floor_data *d);
int orc_floor_apply(const orc_floor *f, orc_floor_data *d, int block_size, float *residue, int reslen);

/* ---------- Residues (Residue0.cs, Residue1.cs, Residue2.cs) ---------- */
typedef struct {
  int type;           /* 0,1,2 */
  int channels;       /* base._channels (1 for type 2) */
  int real_channels;  /* Residue2._channels */
  int begin, end, partition_size, classifications, max_stages;
  int class_book;
  int cascade[64];
  int books[64][8]; /* -1 == null */
  int stages[64];
  int partvals;
  int *decode_map; /* partvals * classbook.dimensions */
} orc_residue;

int orc_residue_init(orc_residue *r, int type, orc_packet *p, int channels, const orc_codebook *books, int nbooks);
void orc_residue_free(orc_residue *r);
int orc_residue_decode(const orc_residue *r, const orc_codebook *books, orc_packet *p,
                       const int *do_not_decode, int nflags, int block_size, float **buffer, int buflen);

/* ---------- Mapping / Mode (Mapping.cs, Mode.cs) ---------- */
typedef struct {
  int coupling_steps;
  int coupling_angle[256], coupling_magnitude[256];
  int submap_count;
  int submap_floor[16], submap_residue[16];
  int channels;
  int channel_floor[256], channel_residue[256]; /* indices into floors/residues */
} orc_mapping;

typedef struct {
  int block_flag, block_size, mapping;
  float *windows[4];
  int ov_start[4], ov_valid[4], ov_total[4];
} orc_mode;

struct orc_decoder {
  /* packet list */
  uint8_t *bytes;
  int64_t *offs;
  int64_t *granule;
  uint8_t *flags;
  int npackets, next_packet;
  /* the Ogg file the list came from (orc_open_ogg): seeking works on its pages */
  uint8_t *ogg_bytes;
  size_t ogg_len;
  int64_t max_granule;

  /* StreamDecoder.cs:19-39 */
  int channels, sample_rate, block0, block1;
  int nbooks, nfloors, nresidues, nmappings, nmodes, mode_field_bits;
  orc_codebook *books;
  orc_floor *floors;
  orc_residue *residues;
  orc_mapping *mappings;
  orc_mode *modes;

  int64_t current_position;
  int has_clipped, has_position, eos_found, clip_samples;
  float **next_buf, **prev_buf; /* [ch][block1] or NULL */
  float **buf_a, **buf_b;       /* backing storage */
  int prev_start, prev_end, prev_stop;

  int last_error;
  /* test hook: the IResidue.Decode calls of the last Mapping.DecodePacket (cursor before the call, residue index) */
  int res_calls, res_call_pos[16], res_call_idx[16], res_call_any;
  /* test hook: the floor data Mapping.DecodePacket worked with for the last packet (after the ForceEnergy / ForceNoEnergy
   * fix-ups), first 8 channels */
  orc_floor_data last_floor[8];
  int last_floor_n;
  int trace_on, trace_n, trace_cap;
  orc_frame_trace *trace;
};

int orc_mapping_init(orc_mapping *m, orc_packet *p, int channels, int nfloors, int nresidues);
int orc_mapping_decode_packet(orc_decoder *d, const orc_mapping *m, orc_packet *p, int block_size, float **buffer);
int orc_mode_init(orc_mode *m, orc_packet *p, int block0, int block1, int nmappings);
void orc_mode_free(orc_mode *m);
/* Mode.Decode: returns 1 decoded, 0 rejected, <0 error */
int orc_mode_decode(orc_decoder *d, const orc_mode *m, orc_packet *p, float **buffer, int *start, int *valid,
                    int *total, int *window_index);

/* Ogg demux -> packet list (orc_ogg.c) */
int orc_ogg_demux(const uint8_t *bytes, size_t len, uint8_t **out_bytes, int64_t **out_offs, int64_t **out_granule,
                  uint8_t **out_flags, int *out_n);

/* largest page granule position of the first logical stream (StreamPageReader._maxGranulePos) */
int orc_ogg_max_granule(const uint8_t *bytes, size_t len, int64_t *max_granule);

#endif
