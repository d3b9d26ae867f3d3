// Known-answer generator: compiles the REFERENCE's packet.h with -DGPU_ON (included from /root/reference, never copied)
// and prints sizeof(Packet) and the byte offset of every member, so that the C-ABI's artis_packet (include/artis_amd.h)
// can be pinned to the struct a reference build would hand over. Also prints the packet_type values this path uses.
// Output: one "name offset size" line per member, then "sizeof <n>", then "enum NAME value" lines.
#include <cstddef>
#include <cstdio>

#include "packet.h"

#define FIELD(f) std::printf("%s %zu %zu\n", #f, offsetof(Packet, f), sizeof(Packet::f))

int main() {
  FIELD(rngstate);
  FIELD(prop_time);
  FIELD(pos);
  FIELD(dir);
  FIELD(nu_cmf);
  FIELD(e_cmf);
  FIELD(nu_rf);
  FIELD(e_rf);
  FIELD(next_trans);
  FIELD(nscatterings);
  FIELD(emissiontype);
  FIELD(em_pos);
  FIELD(em_time);
  FIELD(absorptiontype);
  FIELD(absorptionfreq);
  FIELD(stokes_q);
  FIELD(stokes_u);
  FIELD(trueemissiontype);
  FIELD(trueem_pos);
  FIELD(trueem_time);
  FIELD(type);
  FIELD(cellindex);
  FIELD(escape_type);
  FIELD(escape_time);
  FIELD(tdecay);
  FIELD(number);
  FIELD(originated_from_particlenotgamma);
  FIELD(pellet_decaytype);
  FIELD(pellet_nucindex);
  std::printf("sizeof %zu\n", sizeof(Packet));
  std::printf("enum TYPE_ESCAPE %d\n", static_cast<int>(TYPE_ESCAPE));
  std::printf("enum TYPE_RPKT %d\n", static_cast<int>(TYPE_RPKT));
  std::printf("enum TYPE_KPKT %d\n", static_cast<int>(TYPE_KPKT));
  std::printf("enum TYPE_PRE_KPKT %d\n", static_cast<int>(TYPE_PRE_KPKT));
  std::printf("enum TYPE_RADIOACTIVE_PELLET %d\n", static_cast<int>(TYPE_RADIOACTIVE_PELLET));
  std::printf("enum EMTYPE_NOTSET %d\n", EMTYPE_NOTSET);
  std::printf("enum EMTYPE_FREEFREE %d\n", EMTYPE_FREEFREE);
  std::printf("enum ABSTYPE_FREEFREE %d\n", static_cast<int>(ABSTYPE_FREEFREE));
  std::printf("enum ABSTYPE_BOUNDFREE %d\n", static_cast<int>(ABSTYPE_BOUNDFREE));
  return 0;
}
